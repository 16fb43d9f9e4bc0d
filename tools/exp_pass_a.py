"""conv3 of a layer-3 bottleneck (M=12544, N=1024, K=256, input BatchNorm+ReLU fused): every kernel variant, as the normal
conv (store + integer-atomic statistics), as the statistics-only first pass and as the output-BatchNorm second pass."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
from microbench import time_ops
lib = L.load()
shapes = {"l3": (64, 14, 14, 256, 1024), "l2": (64, 28, 28, 128, 512), "l4": (64, 7, 7, 512, 2048), "l1": (64, 56, 56, 64, 256)}
N, H, W, Cin, Cout = shapes[sys.argv[1] if len(sys.argv) > 1 else "l3"]
M = N * H * W
x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
w = (torch.randn(Cout, Cin, device="cuda") / Cin ** 0.5).bfloat16()
out = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
res = torch.randn(M, Cout, device="cuda").bfloat16()
sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
acc = torch.zeros(2, 1, 2, Cout, dtype=torch.int64, device="cuda")
gam, bet = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")


def op(mode, v):
    o = L.SatOp()
    o.kind, o.dtype = L.OP_CONV, L.SAT_BF16
    o.in0, o.w, o.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, H, W, Cout
    o.KH, o.KW, o.stride, o.pad = 1, 1, 1, 0
    o.sN, o.sH, o.sW = H * W * Cin, W * Cin, Cin
    o.scale0, o.shift0 = sc.data_ptr(), sh.data_ptr()
    o.variant = v
    o.count, o.momentum, o.eps = M, 0.1, 1e-5
    if mode in ("normal", "stats"):
        o.stat_acc, o.stat_shards = acc.data_ptr(), 1
        if mode == "stats":
            o.flags = L.CONV_STATS_ONLY
    elif mode == "outbn":
        o.stat_acc, o.stat_shards = acc.data_ptr(), 1
        o.gamma, o.beta, o.running_mean, o.running_var = gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr()
        o.in1 = res.data_ptr()
        o.flags = 1 | L.CONV_OUT_BN
    return o


print("M=%d N=%d K=%d" % (M, Cout, Cin))
for v in ([3, 12, 32] if os.environ.get("SAT_EXP_FEW") else list(range(1, 22)) + [28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39]):
    row = []
    for mode in ("normal", "stats", "outbn", "plain"):
        o = op(mode, v)
        ops = (L.SatOp * 1)(o)
        try:
            L.check(lib.sat_run_ops(ops, 1, L.stream()))
            torch.cuda.synchronize()
            row.append(time_ops(ops, 1, 20))
        except Exception as e:
            row.append(float("nan"))
    print("variant %2d: conv+stats+store %6.1f  stats-only %6.1f  out-bn+residual %6.1f  store only %6.1f us" % (v, *row))


if Cin == 256 and lib.sat_conv3_fused_ok(M, Cout, Cin):
    acc2 = torch.zeros(2, 1, 2, Cin, dtype=torch.int64, device="cuda")
    acc2[:, 0, 1] = int(M * 2 ** 22)
    sync_w, err = torch.zeros(2, dtype=torch.int32, device="cuda"), torch.zeros(4, dtype=torch.int32, device="cuda")
    o = L.SatOp()
    o.kind, o.dtype = L.OP_CONV3_FUSED, L.SAT_BF16
    o.in0, o.w, o.in1, o.out = x.data_ptr(), w.data_ptr(), res.data_ptr(), out.data_ptr()
    o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, H, W, Cout
    o.KH, o.KW, o.stride, o.pad = 1, 1, 1, 0
    o.stat_acc1, o.stat_shards1, o.gamma1, o.beta1 = acc2.data_ptr(), 1, sc.data_ptr(), sh.data_ptr()
    o.stat_acc, o.stat_shards, o.gamma, o.beta = acc.data_ptr(), 1, gam.data_ptr(), bet.data_ptr()
    o.count, o.momentum, o.eps = M, 0.1, 1e-5
    o.scale_out, o.shift_out = sync_w.data_ptr(), err.data_ptr()
    ops = (L.SatOp * 1)(o)
    L.check(lib.sat_run_ops(ops, 1, L.stream()))
    torch.cuda.synchronize()
    print("SAT_OP_CONV3_FUSED (token acquire + fused launch): %.1f us per call, err %d" % (time_ops(ops, 1, 20), int(err[0])))

    grid = (M + 127) // 128 * (Cout // 512)
    stamps = torch.zeros(grid, 8, dtype=torch.int64, device="cuda")
    lib.sat_conv3_fused_debug(stamps.data_ptr())
    for _ in range(3):
        L.check(lib.sat_run_ops(ops, 1, L.stream()))
    torch.cuda.synchronize()
    lib.sat_conv3_fused_debug(None)
    t = stamps.cpu().double()
    t0 = t[:, 0].min()
    names = ["start", "A landed + bn2 table", "K phase done", "statistics acknowledged", "grid barrier passed", "epilogue done"]
    print("phase stamps (s_memtime ticks of 10 ns; over workgroups, relative to the first workgroup's start):")
    for k in range(6):
        col = (t[:, k] - t0) / 100.0
        print("  %-26s median %7.2f us  min %7.2f  max %7.2f" % (names[k], col.median().item(), col.min().item(), col.max().item()))
