"""Show-Attend-Tell decoder step: host enqueue time vs GPU time (is the step launch-bound on the host?)"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sat = importlib.import_module("show-and-tell_amd")
B, T, V = 64, 20, 10000
torch.manual_seed(123)
model = sat.ShowAttendTellModel(1024, 512, V, 512, None, compute_dtype="bf16").cuda()
images = torch.randn(B, 3, 224, 224, device="cuda")
caps = torch.randint(4, V, (B, T), device="cuda"); caps[:, 0], caps[:, -1] = 1, 2
l1 = [T - 1] * B
targets, _ = sat.pack_targets(caps, [T] * B)
crit = torch.nn.CrossEntropyLoss()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
feats, fmean = model._encode(images)
def phase(name, fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-34s host %.3f ms, total %.3f ms" % (name, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
out = {}
def fwd():
    out["logits"] = model.decode(feats, fmean, caps[:, :-1], l1)
def fwd_bwd():
    model.zero_grad(); loss = crit(model.decode(feats, fmean, caps[:, :-1], l1), targets); loss.backward()
def full():
    fwd_bwd()
    for p in opt.param_groups[0]["params"]: p.grad.data.clamp_(-0.1, 0.1)
    opt.step()
with torch.no_grad():
    phase("decoder forward (no grad)", fwd)
phase("decoder forward+CE+backward", fwd_bwd)
phase("... + clamp + torch Adam", full)
phase("VGG features", lambda: model._encode(images))
