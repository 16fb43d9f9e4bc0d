#!/usr/bin/env python3
"""Throughput of the LEVEL-1 drop-in loop (the reference's own train.py:134-146 body on the HIP-backed
modules -- `features = encoder(images)`, `outputs = decoder(features, captions, lengths)`, torch CrossEntropyLoss, loss.backward(),
then either torch clamp_ + Adam or sat.FusedClampAdam), with and without EncoderCNN.prefetch, next to the fused TrainStep."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sat = importlib.import_module("show-and-tell_amd")
B, T, V = 64, 20, 10000
torch.manual_seed(123)
enc, dec = sat.EncoderCNN(256).cuda().train(), sat.DecoderRNN(256, 512, V, 1).cuda().train()
params = [p for p in list(enc.parameters()) + list(dec.parameters()) if p.requires_grad]
crit = torch.nn.CrossEntropyLoss()
batches = [torch.randn(B, 3, 224, 224, device="cuda") for _ in range(4)]
caps = torch.randint(4, V, (B, T), device="cuda"); caps[:, 0], caps[:, -1] = 1, 2
lengths = [T] * B
targets, l1 = sat.pack_targets(caps, lengths)


def loop(n, opt, fused, prefetch):
    for i in range(n):
        if prefetch:
            for j in range(i + 1, i + 1 + enc.lookahead_depth):
                if j < n:
                    enc.prefetch(batches[j % 4])
        opt.zero_grad()
        loss = crit(dec(enc(batches[i % 4]), caps[:, :-1], l1), targets)
        loss.backward()
        if not fused:
            for p in params:
                p.grad.data.clamp_(-0.1, 0.1)
        opt.step()
    return loss


for fused in (False, True):
    opt = sat.FusedClampAdam(params, lr=1e-3, clip=0.1) if fused else torch.optim.Adam(params, lr=1e-3)
    for prefetch in (False, True):
        loop(6, opt, fused, prefetch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        loss = loop(n, opt, fused, prefetch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("drop-in loop, %s, prefetch %s: %.2f ms/step = %.0f img/s (loss %.3f)"
              % ("FusedClampAdam" if fused else "torch clamp_ + Adam", "on" if prefetch else "off", dt * 1e3, B / dt, loss.item()), flush=True)
