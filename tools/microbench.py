#!/usr/bin/env python3
"""GPU micro-benchmarks of single libsat_hip.so ops at the real ResNet-152 / decoder shapes (tuning aid).
    python tools/microbench.py add        # BN+add+ReLU streams
    python tools/microbench.py conv       # every distinct conv geometry, every variant
"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sat = importlib.import_module("show-and-tell_amd")
L = sat._lib


def time_ops(ops, n, reps=20):
    """us per pass over ops[0:n]; the op list is replicated so ONE C call launches reps passes back to back
    (a Python/ctypes call per launch would make anything shorter than ~7 us look host-bound)."""
    lib = L.load()
    rep = (L.SatOp * (n * reps))(*[ops[i % n] for i in range(n * reps)])
    L.check(lib.sat_run_ops(rep, n * reps, L.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.sat_run_ops(rep, n * reps, L.stream()))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def bench_add(offsets=(0,)):
    for (M, Cc) in [(200704, 256), (50176, 512), (12544, 1024), (3136, 2048)]:
        for off in offsets:
            n = M * Cc
            pool = torch.empty(3 * n + 3 * 65536, dtype=torch.bfloat16, device="cuda").normal_()
            a = pool[0:n]
            b = pool[n + off: 2 * n + off]
            o_ = pool[2 * n + 2 * off: 3 * n + 2 * off]
            sc = torch.rand(Cc, device="cuda") + 0.5
            sh = torch.randn(Cc, device="cuda")
            for kind, name in ((L.OP_BN_ADD_RELU, "add"), (L.OP_BN_RELU, "bnrelu")):
                o = L.SatOp()
                o.kind, o.dtype = kind, L.SAT_BF16
                o.in0, o.in1, o.out = a.data_ptr(), b.data_ptr(), o_.data_ptr()
                o.scale0, o.shift0 = sc.data_ptr(), sh.data_ptr()
                o.N, o.Hout, o.Wout, o.Cout = 1, M, 1, Cc
                us = time_ops(C.pointer(o), 1)
                nbytes = n * 2 * (3 if kind == L.OP_BN_ADD_RELU else 2)
                print("%-7s M=%6d C=%4d off=%6d  %8.1f us  %6.2f TB/s" % (name, M, Cc, off, us, nbytes / us / 1e6))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "add"
    if what == "add":
        bench_add(offsets=(0, 4096 + 256, 32768 + 2048))


def bench_conv(shapes=None, reps=20):
    """layer-3 conv geometries of ResNet-152 at batch 64 (the bulk of the step), autotuned variant"""
    shapes = shapes or [(64, 14, 14, 1024, 256, 1, 1, 0), (64, 14, 14, 256, 256, 3, 1, 1), (64, 14, 14, 256, 1024, 1, 1, 0)]
    lib = L.load()
    for (N, H, W, Cin, Cout, k, stride, pad) in shapes:
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
        w = (torch.randn(Cout, k * k * Cin, device="cuda") / (Cin * k * k) ** 0.5).bfloat16()
        out = torch.empty(N * Ho * Wo, Cout, device="cuda", dtype=torch.bfloat16)
        tiles = lib.sat_conv_tiles_m(N * Ho * Wo)
        part = torch.empty(tiles, 2, Cout, device="cuda")
        o = L.SatOp()
        o.kind, o.dtype = L.OP_CONV, L.SAT_BF16
        o.in0, o.w, o.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
        o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, Ho, Wo, Cout
        o.KH, o.KW, o.stride, o.pad = k, k, stride, pad
        o.sN, o.sH, o.sW = H * W * Cin, W * Cin, Cin
        mode = os.environ.get("SAT_MB_STATS", "partial")      # how the BatchNorm sums leave the kernel: slabs / integer atomics / not at all
        if mode == "partial":
            o.stat_partial, o.tiles_m = part.data_ptr(), tiles
        elif mode == "acc":
            acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = acc.data_ptr()
        ops = (L.SatOp * 1)(o)
        if os.environ.get("SAT_VARIANT"):
            ops[0].variant = int(os.environ["SAT_VARIANT"])
        else:
            scratch = torch.empty(4096, device="cuda")
            L.check(lib.sat_conv_autotune(ops, 1, 5, scratch.data_ptr(), 16384, L.stream()))
        us = time_ops(ops, 1, reps)
        fl = 2.0 * N * Ho * Wo * Cout * k * k * Cin
        print("conv M=%d N=%d K=%d variant %d stats=%s: %.1f us  %.0f TFLOP/s" % (N * Ho * Wo, Cout, k * k * Cin, ops[0].variant, mode, us, fl / us / 1e6))


ALL_SHAPES = [(64, 56, 56, 256, 64, 1, 1, 0), (64, 56, 56, 64, 64, 3, 1, 1), (64, 56, 56, 64, 256, 1, 1, 0),
              (64, 28, 28, 512, 128, 1, 1, 0), (64, 28, 28, 128, 128, 3, 1, 1), (64, 28, 28, 128, 512, 1, 1, 0),
              (64, 14, 14, 1024, 256, 1, 1, 0), (64, 14, 14, 256, 256, 3, 1, 1), (64, 14, 14, 256, 1024, 1, 1, 0),
              (64, 7, 7, 2048, 512, 1, 1, 0), (64, 7, 7, 512, 512, 3, 1, 1), (64, 7, 7, 512, 2048, 1, 1, 0)]

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "conv":
    bench_conv(ALL_SHAPES if len(sys.argv) > 2 and sys.argv[2] == "all" else None)


def bench_conv_inbn(reps=20):
    """conv3 of layer 3 (M=12544, N=1024, K=256) with and without the fused input BatchNorm+ReLU, every variant"""
    lib = L.load()
    N, H, W, Cin, Cout = 64, 14, 14, 256, 1024
    x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, device="cuda") / 16).bfloat16()
    out = torch.empty(N * H * W, Cout, device="cuda", dtype=torch.bfloat16)
    tiles = lib.sat_conv_tiles_m(N * H * W)
    part = torch.empty(tiles, 2, Cout, device="cuda")
    sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
    for fused in (0, 1):
        for v in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
            o = L.SatOp()
            o.kind, o.dtype = L.OP_CONV, L.SAT_BF16
            o.in0, o.w, o.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
            o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, H, W, Cout
            o.KH, o.KW, o.stride, o.pad = 1, 1, 1, 0
            o.sN, o.sH, o.sW = H * W * Cin, W * Cin, Cin
            o.stat_partial, o.tiles_m, o.variant = part.data_ptr(), tiles, v
            if fused:
                o.scale0, o.shift0 = sc.data_ptr(), sh.data_ptr()
            ops = (L.SatOp * 1)(o)
            print("conv3 fused=%d variant %2d: %.1f us" % (fused, v, time_ops(ops, 1, reps)))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "inbn":
    bench_conv_inbn()


def bench_gemm(reps=30):
    """the decoder's f32 GEMMs at cfg2 (rows N=1216, E=256, H=512, V=10000); SAT_GEMM_TILE=1/2/3 forces a tile"""
    lib = L.load()
    N, E, H, V = 1216, 256, 512, 10000
    shapes = [("logits fwd      ", 0, 0, N, V, H, 1), ("dHs = dL.W      ", 0, 1, N, H, V, 3), ("dW  = dL^T.Hs   ", 2, 1, V, H, N, 1),
              ("xg  = X.Wih^T   ", 0, 0, N, 4 * H, E, 1), ("dWih = DG^T.X   ", 2, 1, 4 * H, E, N, 1),
              ("dWhh = DG^T.HP  ", 2, 1, 4 * H, H, N, 1), ("dX  = DG.Wih    ", 0, 1, N, E, 4 * H, 1)]
    for name, am, bm, M, Nn, K, ks in shapes:
        A = torch.randn((M, K) if am == 0 else (K, M), device="cuda")
        B = torch.randn((Nn, K) if bm == 0 else (K, Nn), device="cuda")
        Cm = torch.empty(ks, M, Nn, device="cuda")
        def run():
            if ks == 1:
                L.check(lib.sat_gemm_f32(am, bm, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), Cm.data_ptr(), Nn, None, None,
                                         M, Nn, K, L.stream()))
            else:
                L.check(lib.sat_gemm_f32_splitk(am, bm, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), Cm.data_ptr(), Nn,
                                                None, None, M, Nn, K, ks, M * Nn, L.stream()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print("%s M=%5d N=%5d K=%5d ks=%d: %7.1f us  %5.1f TFLOP/s" % (name, M, Nn, K, ks, us, 2.0 * M * Nn * K / us / 1e6))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "gemm":
    bench_gemm()


def bench_gemm_bf16(reps=30):
    """the bf16-mode vocab GEMMs at cfg2 (sat_gemm_bf16_nt); SAT_GEMM_BF16_S=2/3/4 forces the ring depth"""
    lib = L.load()
    N, H, V = 1216, 512, 10000
    Vp = (V + 63) // 64 * 64
    shapes = [("logits = Hs W^T   ", N, V, H, 1), ("dW = G^T Hs       ", V, H, N, 1), ("dHs = G W (ks=6)  ", N, H, Vp, 6), ("dHs = G W (ks=12) ", N, H, Vp, 12)]
    for name, M, Nn, K, ks in shapes:
        A = torch.randn(M, K, device="cuda").bfloat16()
        B = torch.randn(Nn, K, device="cuda").bfloat16()
        Cm = torch.empty(ks, M, Nn, device="cuda")
        def run():
            L.check(lib.sat_gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, Cm.data_ptr(), Nn, None, M, Nn, K, ks, M * Nn, L.stream()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print("S=%s %s M=%5d N=%5d K=%5d ks=%2d: %7.1f us  %6.1f TFLOP/s" % (os.environ.get("SAT_GEMM_BF16_S", "auto"), name, M, Nn, K, ks, us, 2.0 * M * Nn * K / us / 1e6))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "gemm16":
    bench_gemm_bf16()


def bench_quant(reps=20):
    """Workgroup-count quantisation of the layer-3 convs: the launch time as a function of the batch (rows = batch x 196), with the
    variants the grouped program runs (ring 128x128 two per CU, conv_pr_kernel, conv_xp_kernel 4 column tiles).  If the time is a
    step function of workgroups / CUs, a 128-row tile wastes 23 % of the chip at M = 12544 x G (196 G workgroups per 256 CUs)."""
    lib = L.load()
    geoms = [(14, 14, 1024, 256, 1, 0, 3, 0), (14, 14, 1024, 256, 1, 0, 33, 0), (14, 14, 1024, 256, 1, 0, 34, 0), (14, 14, 256, 256, 3, 1, 30, 1), (14, 14, 256, 256, 3, 1, 32, 1),
             (28, 28, 128, 128, 3, 1, 30, 1), (28, 28, 128, 128, 3, 1, 32, 1), (14, 14, 256, 1024, 1, 0, 29, 1), (14, 14, 256, 1024, 1, 0, 33, 1)]
    if len(sys.argv) > 2:
        geoms = [g for g in geoms if str(g[6]) in sys.argv[2].split(",")]
    for (H, W, Cin, Cout, k, pad, variant, fused) in geoms:
        for N in (32, 42, 43, 64, 83, 84, 96, 128, 166, 168, 192, 250, 256):
            if H == 28:
                N = max(N // 4, 1)
            x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
            w = (torch.randn(Cout, k * k * Cin, device="cuda") / (Cin * k * k) ** 0.5).bfloat16()
            out = torch.empty(N * H * W, Cout, device="cuda", dtype=torch.bfloat16)
            acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
            o = L.SatOp()
            o.kind, o.dtype = L.OP_CONV, L.SAT_BF16
            o.in0, o.w, o.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
            o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, H, W, Cout
            o.KH, o.KW, o.stride, o.pad = k, k, 1, pad
            o.sN, o.sH, o.sW = H * W * Cin, W * Cin, Cin
            o.stat_acc, o.variant = acc.data_ptr(), variant
            wp = torch.empty_like(w)
            L.check(lib.sat_conv_pack_weights(w.data_ptr(), wp.data_ptr(), Cout, Cin, k * k, L.stream()))
            o.w_packed = wp.data_ptr()
            if fused:
                o.scale0, o.shift0 = sc.data_ptr(), sh.data_ptr()
            if os.environ.get("SAT_MB_EPILOGUE") == "1":      # the inference epilogue: affine + residual + ReLU instead of statistics
                osc, osh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
                res = torch.randn(N * H * W, Cout, device="cuda").bfloat16()
                o.stat_acc = None
                o.scale1, o.shift1, o.in1, o.flags = osc.data_ptr(), osh.data_ptr(), res.data_ptr(), 1
            ops = (L.SatOp * 1)(o)
            us = time_ops(ops, 1, reps)
            M = N * H * W
            tiles = -(-M // 128)
            fl = 2.0 * M * Cout * k * k * Cin
            print("K=%4d N=%4d variant %2d batch %3d: M=%6d = %5.1f row tiles of 128 (%.2f x 256)  %6.1f us  %4.0f TFLOP/s  %.3f us per row tile" %
                  (k * k * Cin, Cout, variant, N, M, M / 128, tiles / 256, us, fl / us / 1e6, us / tiles))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "quant":
    bench_quant()

