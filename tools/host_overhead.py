#!/usr/bin/env python3
"""How long the HOST needs to enqueue one training step (launch-bound check: with the encoder look-ahead a step is 4.5 ms of GPU
time).  Measured with the queue EMPTY (two steps after a synchronize, repeated), so back-pressure from a full launch queue does
not hide in the figure; plus a cProfile of the enqueue path."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sat = importlib.import_module("show-and-tell_amd")
torch.manual_seed(123)
model = sat.ShowAndTell(256, 512, 10000, 1).cuda().train()
ts = sat.TrainStep(model)
depth = model.encoder.lookahead_depth
batches = [torch.randn(64, 3, 224, 224, device="cuda") for _ in range(depth + 1)]
caps = torch.randint(4, 10000, (64, 20), device="cuda"); caps[:, 0], caps[:, 19] = 1, 2
lengths = [20] * 64


def run(n, la):
    for i in range(n):
        nxt = [batches[j % (depth + 1)] for j in range(i + 1, i + 1 + depth)] if la else None
        ts.step(batches[i % (depth + 1)], caps, lengths, next_images=nxt)


for la in (False, True):
    run(8, la)
    ts.drop_lookahead()
    torch.cuda.synchronize()
    hs = []
    for trial in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(2, la)
        t1 = time.perf_counter()
        ts.drop_lookahead()
        torch.cuda.synchronize()
        hs.append((t1 - t0) / 2 * 1e3)
    print("lookahead=%d: host enqueue per step with an empty queue: %s ms (median %.3f)" % (la, ["%.3f" % h for h in hs], sorted(hs)[len(hs) // 2]))
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
run(4, True)
pr.disable()
ts.drop_lookahead()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
