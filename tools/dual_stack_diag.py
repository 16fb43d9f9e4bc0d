"""Throughput of TWO frozen conv stacks (two batches) in flight on two streams vs one after the other."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sat = importlib.import_module("show-and-tell_amd")
CFG = bench.CFG
dev = torch.device("cuda", 0)
torch.manual_seed(123)
ms = [sat.ShowAndTell(CFG["embed"], CFG["hidden"], CFG["vocab"], CFG["layers"], compute_dtype="bf16").to(dev).train() for _ in range(3)]
imgs = [bench.synth_batch(torch, CFG["batch"], CFG["vocab"], CFG["cap_len"], CFG["image"], dev, 123 + i)[0] for i in range(3)]
for m, x in zip(ms, imgs):
    for _ in range(3): m.encoder._pooled_raw(x)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    for m, x in zip(ms[:2], imgs): m.encoder._pooled_raw(x)
torch.cuda.synchronize(); seq = (time.perf_counter() - t0) / (2 * n) * 1e3
for k in (2, 3):
    streams = [torch.cuda.Stream() for _ in range(k)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        for j, (s, m, x) in enumerate(zip(streams, ms, imgs)):
            with torch.cuda.stream(s):
                m.encoder._pooled_raw(x)
    torch.cuda.synchronize(); par = (time.perf_counter() - t0) / (k * n) * 1e3
    print("conv stack per batch: sequential %.3f ms, %d in flight %.3f ms" % (seq, k, par))
