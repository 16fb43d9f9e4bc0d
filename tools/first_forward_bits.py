#!/usr/bin/env python3
"""The bits of a first forward, for comparing PROCESSES: seed 123 (config.py:15), BASELINE configs[1] (ResNet-152, batch 64,
embed 256 / hidden 512 / vocab 10000, bf16), the synthetic batch of bench.py.  Prints one JSON line: the mean CE of the first
forward (float.hex: every bit), a SHA-256 of the pooled features of the batch as the ungrouped program computes them and as the
grouped look-ahead program computes them, and the loss after three whole train steps with the look-ahead on.
`tests/test_gpu_reproducible.py` runs this twice and asserts the lines are equal.
    python tools/first_forward_bits.py [batch]"""
import hashlib
import importlib
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

sat = importlib.import_module("show-and-tell_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
torch.manual_seed(123)
model = sat.ShowAndTell(256, 512, 10000, 1, compute_dtype="bf16").to(dev).train()
ts = sat.TrainStep(model, lr=1e-3, grad_clip=0.1)
images, caps, lengths = bench.synth_batch(torch, B, 10000, 20, 224, dev, 123)
others = [bench.synth_batch(torch, B, 10000, 20, 224, dev, 977 * (k + 1))[0] for k in range(2)]


def sha(t):
    return hashlib.sha256(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


with torch.no_grad():
    pooled_single = sha(model.encoder.pooled_features(images))
    model.encoder.prefetch_many([images, others[0]])            # one grouped program run: two batches per launch
    pooled_grouped = sha(model.encoder.pooled_features(images))
    model.encoder.drop_lookahead()
inv = 1.0 / sum(l - 1 for l in lengths)
ce = float(ts.forward_backward((images, caps, lengths), inv).item())
batches = [images] + others
loss = None
for i in range(3):
    nxt = [batches[j % 3] for j in range(i + 1, i + 3) if j < 3]
    loss = ts.step(batches[i % 3], caps, lengths, next_images=nxt or None)
ts.check_ids()
print(json.dumps({"batch": B, "ce_first_forward": ce.hex(), "ce": round(ce, 6), "pooled_single": pooled_single,
                  "pooled_grouped": pooled_grouped, "loss_after_3_steps": float(loss.item()).hex()}), flush=True)
