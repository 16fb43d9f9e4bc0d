#!/usr/bin/env python3
"""Calibration only (never on the product path): what the vendor GEMM library reaches on the layer-3 conv shapes
viewed as plain GEMMs -- a known-good reference for the ceiling of these shapes on this chip."""
import torch
shapes = [(12544, 256, 1024), (12544, 256, 2304), (12544, 1024, 256), (50176, 128, 1152), (200704, 64, 576), (3136, 512, 4608)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("hipBLASLt/rocBLAS bf16 GEMM M=%d N=%d K=%d: %.1f us  %.0f TFLOP/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
