#!/usr/bin/env python3
"""How long the HOST needs to enqueue one training step (launch-bound check for multi-rank runs)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sat = importlib.import_module("show-and-tell_amd")
torch.manual_seed(123)
model = sat.ShowAndTell(256, 512, 10000, 1).cuda().train()
ts = sat.TrainStep(model)
images = torch.randn(64, 3, 224, 224, device="cuda")
caps = torch.randint(4, 10000, (64, 20), device="cuda"); caps[:, 0], caps[:, 19] = 1, 2
lengths = [20] * 64
for _ in range(5):
    ts.step(images, caps, lengths)
torch.cuda.synchronize()
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ts.step(images, caps, lengths)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
