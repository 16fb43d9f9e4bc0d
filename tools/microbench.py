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
    lib = L.load()
    L.check(lib.sat_run_ops(ops, n, L.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.sat_run_ops(ops, n, L.stream()))
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
