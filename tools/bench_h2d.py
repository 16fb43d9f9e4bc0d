#!/usr/bin/env python3
"""PCIe-inclusive rate of the training step (never bench.py's `value`): batches start in pinned HOST memory.
   a) blocking .cuda() per step   b) sat.DevicePrefetcher (copy of batch i+1 on a side stream under step i)"""
import importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sat = importlib.import_module("show-and-tell_amd")
torch.manual_seed(123)
model = sat.ShowAndTell(256, 512, 10000, 1).cuda().train()
ts = sat.TrainStep(model)
n = 30
host = []
for i in range(4):
    im = torch.randn(64, 3, 224, 224).pin_memory()
    cp = torch.randint(4, 10000, (64, 20)); cp[:, 0], cp[:, 19] = 1, 2
    host.append((im, cp.pin_memory(), [20] * 64))
batches = [host[i % 4] for i in range(n)]
dev = [(im.cuda(), cp.cuda(), ln) for im, cp, ln in host]
for i in range(6):
    ts.step(*dev[i % 4])
torch.cuda.synchronize()
out = {}
t0 = time.perf_counter()
for i in range(n):
    ts.step(*dev[i % 4])
torch.cuda.synchronize()
out["resident_img_s"] = round(64 * n / (time.perf_counter() - t0), 1)
t0 = time.perf_counter()
for im, cp, ln in batches:
    ts.step(im.cuda(), cp.cuda(), ln)
torch.cuda.synchronize()
out["blocking_copy_img_s"] = round(64 * n / (time.perf_counter() - t0), 1)
t0 = time.perf_counter()
for im, cp, ln in sat.DevicePrefetcher(batches, "cuda"):
    ts.step(im, cp, ln)
torch.cuda.synchronize()
out["prefetcher_img_s"] = round(64 * n / (time.perf_counter() - t0), 1)
print(json.dumps(out))
