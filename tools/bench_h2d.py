#!/usr/bin/env python3
"""PCIe-inclusive rate of the training step (never bench.py's `value`): batches start in pinned HOST memory.
sequential / look-ahead on resident batches, blocking .cuda() per step, sat.DevicePrefetcher (copies on a side stream), and the
prefetcher at look-ahead depth feeding `next_images` (its upcoming device tensors are announced to the step)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sat = importlib.import_module("show-and-tell_amd")
torch.manual_seed(123)
model = sat.ShowAndTell(256, 512, 10000, 1).cuda().train()
ts = sat.TrainStep(model)
host, dev = [], []
for i in range(4):
    im = torch.randn(64, 3, 224, 224).pin_memory()
    cp = torch.randint(4, 10000, (64, 20)); cp[:, 0], cp[:, 19] = 1, 2
    host.append((im, cp.pin_memory(), [20] * 64))
    dev.append((im.cuda(), cp.cuda(), [20] * 64))
depth = model.encoder.lookahead_depth
def resident(m, la=True):
    for i in range(m):
        nxt = [dev[j % 4][0] for j in range(i + 1, i + 1 + depth) if j < m] if la else None
        ts.step(*dev[i % 4], next_images=nxt or None)
def timed(name, fn):
    fn(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(30); torch.cuda.synchronize()
    print("%s: %.0f img/s" % (name, 64 * 30 / (time.perf_counter() - t0)), flush=True)
timed("sequential first", lambda m: resident(m, False))
timed("look-ahead after sequential", resident)
timed("look-ahead again", resident)
def blocking(m):
    for i in range(m):
        im, cp, ln = host[i % 4]
        ts.step(im.cuda(), cp.cuda(), ln)
timed("blocking copies, sequential", blocking)
timed("look-ahead after blocking copies", resident)
def pref(m):
    for im, cp, ln in sat.DevicePrefetcher([host[i % 4] for i in range(m)], "cuda"):
        ts.step(im, cp, ln)
timed("prefetcher depth 1, sequential", pref)
timed("look-ahead after the prefetcher", resident)
def pref_la(m):
    pf = sat.DevicePrefetcher([host[i % 4] for i in range(m)], "cuda", depth=depth)
    for im, cp, ln in pf:
        ts.step(im, cp, ln, next_images=pf.upcoming_images() or None)
timed("prefetcher depth 3 + look-ahead", pref_la)
