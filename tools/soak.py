#!/usr/bin/env python3
"""Soak: N training steps at cfg2 on varying synthetic batches (ragged lengths); checks finite loss, stable memory."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sat = importlib.import_module("show-and-tell_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(123)
model = sat.ShowAndTell(256, 512, 10000, 1).cuda().train()
ts = sat.TrainStep(model)
g = torch.Generator().manual_seed(5)
batches = []
for i in range(8):
    lengths = sorted(torch.randint(8, 21, (64,), generator=g).tolist(), reverse=True)
    caps = torch.zeros(64, 20, dtype=torch.long)
    for b, l in enumerate(lengths):
        caps[b, 0] = 1
        caps[b, 1:l - 1] = torch.randint(4, 10000, (l - 2,), generator=g)
        caps[b, l - 1] = 2
    batches.append((torch.randn(64, 3, 224, 224, generator=g).cuda(), caps.cuda(), lengths))
mem0 = None
t0 = time.time()
for it in range(n):
    nxt = [batches[j % 8][0] for j in range(it + 1, it + 1 + model.encoder.lookahead_depth) if j < n] if os.environ.get("SAT_LOOKAHEAD", "1") != "0" else None
    loss = ts.step(*batches[it % 8], lr=sat.lr_for_epoch(1 + it // 100), next_images=nxt or None)
    if it % 50 == 49 or it == n - 1:
        torch.cuda.synchronize()
        m = torch.cuda.memory_allocated() / 2**20
        r = torch.cuda.memory_reserved() / 2**20
        if mem0 is None:
            mem0 = r
        print("step %4d loss %.4f  allocated %.0f MiB reserved %.0f MiB  %.1f img/s" % (it + 1, loss.item(), m, r, 64 * (it + 1) / (time.time() - t0)), flush=True)
        assert torch.isfinite(loss).all()
# (the last `lookahead_depth` steps may build the single-batch look-ahead programs the end of the data needs: one-off allocations)
assert torch.cuda.memory_reserved() / 2**20 <= mem0 * 1.05 + 64 + 3 * 900, "memory grew"
for name, p in model.named_parameters():
    assert torch.isfinite(p).all(), name
ts.check_ids()
print("soak ok")
