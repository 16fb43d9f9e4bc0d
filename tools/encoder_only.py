#!/usr/bin/env python3
"""How much of the look-ahead step is the encoder pipeline alone?  The frozen conv stacks of cfg 2 with the default look-ahead
(grouped programs on side streams) and NOTHING on the main stream but taking the pooled features, against bench.py's full step.
    python tools/encoder_only.py [batches]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sat = importlib.import_module("show-and-tell_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(123)
model = sat.ShowAndTell(256, 512, 10000, 1, compute_dtype="bf16").cuda().train()
enc = model.encoder
g = torch.Generator().manual_seed(5)
batches = [torch.randn(64, 3, 224, 224, generator=g).cuda() for _ in range(8)]
enc.build_lookahead(batches[0])
depth = enc.lookahead_depth


def run(k):
    with torch.no_grad():
        for i in range(k):
            enc.prefetch_many([batches[j % 8] for j in range(i + 1, i + 1 + depth) if j < k])
            enc.pooled_features(batches[i % 8])


run(12)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("encoder pipeline alone, %d batches: %.3f ms per batch = %.0f img/s (groups %d, depth %d, %d streams)"
          % (n, dt / n * 1e3, 64 * n / dt, enc.lookahead_groups, depth, enc.lookahead_streams or enc._n_slots()), flush=True)
