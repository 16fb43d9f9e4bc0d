#!/usr/bin/env python3
"""Decode loops alone (no encoder): `DecoderRNN.sample_beam` / `sample` at BASELINE configs[4] shapes (64 images, beam 5, embed 256,
hidden 512, vocab 10000), for `rocprofv3 --kernel-trace --stats` (per-kernel time of the 20-step loops) and wall time.
    python tools/decode_trace.py [beam] [reps]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

sat = importlib.import_module("show-and-tell_amd")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(123)
dec = sat.DecoderRNN(256, 512, 10000, 1).cuda().eval()
feats = torch.randn(64, 256, device="cuda")
fn = (lambda: dec.sample_beam(feats, K, end_id=2)) if K > 1 else (lambda: dec.sample(feats))
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("beam %d: %.3f ms per 20-step decode of 64 images (%d reps)" % (K, (time.perf_counter() - t0) / reps * 1e3, reps), flush=True)
