#!/usr/bin/env python3
"""Throughput of the OTHER BASELINE.json configurations on one MI355X (bench.py is configs[1]):
   configs[3]  Inception-v3 encoder (299x299) + 2-layer LSTM hidden 1024 (embed 512), batch 64, bf16 conv stack: train step img/s
    python tools/bench_configs.py"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sat = importlib.import_module("show-and-tell_amd")
INCEPTION_V3_CONV_MACS = 5711168096        # per 299x299 image, 94 convs (oracle.inception.conv_macs(); tests/test_oracle_inception_macs below keeps them equal)

B, T, V = 64, 20, 10000
torch.manual_seed(123)
model = sat.ShowAndTell(512, 1024, V, 2, arch="inception_v3", compute_dtype="bf16").cuda().train()
ts = sat.TrainStep(model)
images = torch.randn(B, 3, 299, 299, device="cuda")
caps = torch.randint(4, V, (B, T), device="cuda")
caps[:, 0], caps[:, -1] = 1, 2
lengths = [T] * B
DEPTH = model.encoder.lookahead_depth
NB = DEPTH + 1
batches = [images] + [torch.randn(B, 3, 299, 299, device="cuda") for _ in range(NB - 1)]
LOOKAHEAD = os.environ.get("SAT_LOOKAHEAD", "1") != "0"


def run(n):
    out = None
    for i in range(n):
        nxt = [batches[j % NB] for j in range(i + 1, i + 1 + DEPTH) if j < n] if LOOKAHEAD else None
        out = ts.step(batches[i % NB], caps, lengths, next_images=nxt or None)
    return out


loss = run(6)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
loss = run(n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
prog = model.encoder._program(images)
prog.run_timed(images)
_, us = prog.run_timed(images)
gf = 2.0 * INCEPTION_V3_CONV_MACS * B / 1e9
print("configs[3] Inception-v3 299x299 + L=2 H=1024 E=512, batch 64, bf16 (encoder look-ahead %s): %.2f ms/step = %.0f img/s (loss %.4f); conv launches %d, "
      "%.2f ms in conv kernels = %.0f TFLOP/s (%.3f of the 2.5 PFLOP/s bf16 peak)"
      % ("on" if LOOKAHEAD else "off", dt * 1e3, B / dt, loss.item(), len(us), sum(us) * 1e-3, gf / (sum(us) * 1e-6) / 1e3, gf / (sum(us) * 1e-6) / 1e3 / 2500))
