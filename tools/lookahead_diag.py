"""Where the look-ahead step goes: host enqueue time vs GPU time, conv stack alone, decoder alone (cached pooled features)."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sat = importlib.import_module("show-and-tell_amd")
CFG = bench.CFG
dev = torch.device("cuda", 0)
torch.manual_seed(123)
model = sat.ShowAndTell(CFG["embed"], CFG["hidden"], CFG["vocab"], CFG["layers"], compute_dtype="bf16").to(dev).train()
ts = sat.TrainStep(model)
images, caps, lengths = bench.synth_batch(torch, CFG["batch"], CFG["vocab"], CFG["cap_len"], CFG["image"], dev, 123)
images_b = bench.synth_batch(torch, CFG["batch"], CFG["vocab"], CFG["cap_len"], CFG["image"], dev, 977)[0]
bt = [images, images_b]
def run(n, la):
    for i in range(n):
        nxt = bt[(i + 1) & 1] if (la and i + 1 < n) else None
        ts.step(bt[i & 1], caps, lengths, next_images=nxt)
for la in (False, True, False, True):
    run(5, la); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(20, la); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("lookahead=%d host enqueue %.3f ms/step, total %.3f ms/step" % (la, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
enc = model.encoder
for _ in range(3): enc._pooled_raw(images)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): enc._pooled_raw(images)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("conv stack alone: host %.3f, total %.3f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
feats = torch.randn(CFG["batch"], CFG["embed"], device=dev)
for _ in range(3): ts.step(feats, caps, lengths)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ts.step(feats, caps, lengths)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("decoder-only step (cached features): host %.3f, total %.3f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
