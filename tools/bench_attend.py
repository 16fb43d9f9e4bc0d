#!/usr/bin/env python3
"""Throughput of the Show-Attend-Tell path (model2.py, the model train.py:37 constructs) on one MI355X: training step
(VGG16 features[:-3] frozen + attention decoder fwd + CE + hand-written backward + clamp + torch Adam, train.py:134-146) and
greedy sampling (eval.py:99), batch 64, 224x224, hidden 1024 / embed 512 (config.py:27-28 defaults), vocab 10000, len-20 captions.
    python tools/bench_attend.py [bf16|f32]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sat = importlib.import_module("show-and-tell_amd")
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, T, V = 64, 20, 10000
torch.manual_seed(123)
model = sat.ShowAttendTellModel(1024, 512, V, 512, None, compute_dtype=dtype).cuda()
images = torch.randn(B, 3, 224, 224, device="cuda")
caps = torch.randint(4, V, (B, T), device="cuda")
caps[:, 0], caps[:, -1] = 1, 2
lengths = [T] * B
targets, l1 = sat.pack_targets(caps, lengths)
FUSED_OPT = os.environ.get("SAT_FUSED_OPT", "1") != "0"     # clip_gradient + Adam as one launch (sat.FusedClampAdam) vs torch
if FUSED_OPT:
    opt = sat.FusedClampAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, clip=0.1)
else:
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
crit = torch.nn.CrossEntropyLoss()


images_b = torch.randn(B, 3, 224, 224, device="cuda")
batches = [images, images_b]
LOOKAHEAD = os.environ.get("SAT_LOOKAHEAD", "1") != "0"


def step(i=0, last=True):
    if FUSED_OPT:
        opt.zero_grad()
    else:
        model.zero_grad()
    x = batches[i & 1]
    # forward of THIS batch first consumes its (possibly prefetched) features; the NEXT batch's frozen VGG stack then starts on
    # the side stream and runs under this batch's decoder forward / backward / Adam (ShowAttendTellModel.prefetch_features)
    feats, fmean = model._encode(x)
    if LOOKAHEAD and not last:
        model.prefetch_features(batches[(i + 1) & 1])
    loss = crit(model.decode(feats, fmean, caps[:, :-1], l1), targets)
    loss.backward()
    if not FUSED_OPT:
        for p in opt.param_groups[0]["params"]:
            p.grad.data.clamp_(-0.1, 0.1)
    opt.step()
    return loss


for i in range(3):
    loss = step(i, i == 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for i in range(n):
    loss = step(i, i == n - 1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("Show-Attend-Tell train step (%s conv stack, drop-in autograd path + torch CE, %s): %.2f ms/step = %.0f img/s, loss %.4f"
      % (dtype, "FusedClampAdam" if FUSED_OPT else "torch clamp + Adam", dt * 1e3, B / dt, loss.item()))
with torch.no_grad():
    feats, fmean = model._encode(images)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model._encode(images)
    torch.cuda.synchronize()
    enc = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        ids = model.sample_features(feats, None)
    torch.cuda.synchronize()
    dec = (time.perf_counter() - t0) / 5
print("VGG16 features[:-3] forward %.2f ms (%.0f TFLOP/s); greedy sample (20 steps) %.2f ms => %.0f captions/s end to end"
      % (enc * 1e3, 2 * 14.884e9 * B / enc / 1e12, dec * 1e3, B / (enc + dec)))
with torch.no_grad():          # eval.py:99 as a loop over batches with the next batch's frozen VGG stack prefetched
    def pipeline(n):
        for i in range(n):
            if i + 1 < n:
                model.prefetch_features(batches[(i + 1) & 1])
            model.sample(batches[i & 1])
    pipeline(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 12
    pipeline(n)
    torch.cuda.synchronize()
    print("greedy sample with the features look-ahead: %.0f captions/s end to end" % (B * n / (time.perf_counter() - t0)))
    hp = torch.cuda.Stream(priority=-1)          # the decode loop's small dependent launches ahead of the conv workgroups
    hp.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(hp):
        pipeline(3)
        hp.synchronize()
        t0 = time.perf_counter()
        pipeline(n)
        hp.synchronize()
        print("... the same on a high-priority stream: %.0f captions/s" % (B * n / (time.perf_counter() - t0)))
