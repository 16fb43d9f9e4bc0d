#!/usr/bin/env python3
"""Timeline of ONE timed region of bench.py's training loop (cfg 2, look-ahead): when every conv-stack program starts and ends on
its side stream, when the main stream gets each batch's pooled features and when each step's decoder work ends -- HIP events, times in
ms from the region's start.  Shows where a 20-step region (the contract's) spends its fill and drain.
    python tools/step_timeline.py [steps] [inception]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sat = importlib.import_module("show-and-tell_amd")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(123)
dev = torch.device("cuda", 0)
INC = len(sys.argv) > 2 and sys.argv[2].startswith("inc")          # BASELINE configs[3]: Inception-v3 299 x 299 + L = 2 / H = 1024
model = (sat.ShowAndTell(512, 1024, 10000, 2, arch="inception_v3", compute_dtype="bf16") if INC else
         sat.ShowAndTell(256, 512, 10000, 1, compute_dtype="bf16")).to(dev).train()
ts = sat.TrainStep(model, lr=1e-3)
enc = model.encoder
g = torch.Generator().manual_seed(5)
depth = enc.lookahead_depth
nb = depth + 1
batches = [torch.randn(64, 3, 299 if INC else 224, 299 if INC else 224, generator=g).to(dev) for _ in range(nb)]
lengths = [20] * 64
caps = torch.randint(3, 10000, (64, 20), generator=g).to(dev)
enc.build_lookahead(batches[0])

rec = {"progs": [], "handed": [], "done": []}
ev = lambda: torch.cuda.Event(enable_timing=True)
orig_launch = enc._launch
orig_pooled = ts._encoder_pooled
ids = {}


def launch(ims, slot, groups):
    dv = ims[0].device
    stream = enc._slot_stream(dv, slot)
    e0 = ev()
    stream.wait_stream(torch.cuda.current_stream(dv))
    e0.record(stream)
    orig_launch(ims, slot, groups)
    e1 = ev()
    e1.record(stream)
    rec["progs"].append((e0, e1, slot, [ids.get(id(im), "?") for im in ims]))


def pooled(images, out, next_images=None):
    hit = any(images is im for f in enc._inflight for im in f["images"])
    r = orig_pooled(images, out, next_images)
    e = ev()
    e.record()
    rec["handed"].append((e, hit))
    return r


enc._launch = launch
ts._encoder_pooled = pooled


def run(n, mark=False):
    for i in range(n):
        for j in range(nb):
            ids[id(batches[(i + j) % nb])] = i + j if j <= depth else "?"
        nxt = [batches[j % nb] for j in range(i + 1, i + 1 + depth) if j < n]
        ts.step(batches[i % nb], caps, lengths, next_images=nxt or None)
        if mark:
            e = ev()
            e.record()
            rec["done"].append(e)


run(5)
torch.cuda.synchronize()
for rep in range(2):
    for k in rec:
        rec[k].clear()
    t0 = ev()
    t0.record()
    run(K, True)
    tend = ev()
    tend.record()
    torch.cuda.synchronize()
print("region of %d steps: %.3f ms = %.3f ms/step" % (K, t0.elapsed_time(tend), t0.elapsed_time(tend) / K))
print("step: pooled features handed to the main stream at / decoder + optimizer done at   (ms from the region's start)")
for i, ((eh, hit), ed) in enumerate(zip(rec["handed"], rec["done"])):
    print("  step %2d  features %7.3f (%s)   done %7.3f   decoder+optimizer %.3f" % (
        i, t0.elapsed_time(eh), "look-ahead" if hit else "own stack on the main stream", t0.elapsed_time(ed), eh.elapsed_time(ed)))
print("conv-stack programs on the side streams: batches, slot, start -> end (duration)")
for e0, e1, slot, bs in rec["progs"]:
    print("  batches %-10s slot %d  %7.3f -> %7.3f  (%.3f)" % (bs, slot, t0.elapsed_time(e0), t0.elapsed_time(e1), e0.elapsed_time(e1)))
