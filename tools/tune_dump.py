"""Print the conv autotuner's per-variant timings (SAT_TUNE_VERBOSE) for the ResNet-152 program at batch 64: the ungrouped
program and the grouped one of the look-ahead (G batches per launch; `python tools/tune_dump.py [G]`, default 2), then every conv
launch of one in-order pass of each with its duration.  `python tools/tune_dump.py 3 inception`: the Inception-v3 program of BASELINE
configs[3] at 299 x 299 instead."""
import importlib, os, sys
os.environ["SAT_TUNE_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(1)
if len(sys.argv) > 2 and sys.argv[2].startswith("inc"):
    m = sat.ShowAndTell(512, 1024, 10000, 2, arch="inception_v3", compute_dtype="bf16").cuda().train()
    x = torch.randn(64, 3, 299, 299, device="cuda")
else:
    m = sat.ShowAndTell(256, 512, 10000, 1, compute_dtype="bf16").cuda().train()
    x = torch.randn(64, 3, 224, 224, device="cuda")
enc = m.encoder
p1 = enc._program(x)
pg = enc._program(x, instance="g0", groups=G)
torch.cuda.synchronize()
for name, prog, arg, g in (("one batch per launch", p1, x, 1), ("%d batches per launch" % G, pg, [x] * G, G)):
    prog.run_timed(arg)
    _, us = prog.run_timed(arg)
    idx = [i for i in range(prog.n_ops) if prog.ops[i].kind == L.OP_CONV]
    agg = {}
    for k, i in enumerate(idx):
        o = prog.ops[i]
        key = (o.N * o.Hout * o.Wout, o.Cout, o.KH * o.KW * o.Cin, o.stride, int(o.variant), 1 if (o.scale0 or o.stat_acc1) else 0, "%dx%d" % (o.KH, o.KW))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += us[k]
    print("== %s: %d conv launches, %.3f ms in sequence (%.3f ms per batch)" % (name, len(idx), sum(us) * 1e-3, sum(us) * 1e-3 / g), file=sys.stderr)
    for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        M, N, K, s, v, fin, kk = key
        fl = 2.0 * M * N * K * g
        print("   M=%6d N=%4d K=%4d %s stride %d v%-2d%s  n=%2d  avg %6.1f us  total %6.3f ms  %4.0f TFLOP/s" %
              (M, N, K, kk, s, v, " +inBN" if fin else "      ", n, t / n, t * 1e-3, fl / (t / n) / 1e6), file=sys.stderr)
