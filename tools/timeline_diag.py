"""Timeline of the look-ahead training step (un-profiled: rocprofv3 serialises the streams).  HIP events on the main stream at the
phase boundaries of each step and on the side streams around each prefetched conv stack; prints per-step phase spans and the
stacks' start/end relative to the first event.   python tools/timeline_diag.py [steps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sat = importlib.import_module("show-and-tell_amd")
M = importlib.import_module("show-and-tell_amd.models")
CFG = bench.CFG
dev = torch.device("cuda", 0)
torch.manual_seed(123)
model = sat.ShowAndTell(CFG["embed"], CFG["hidden"], CFG["vocab"], CFG["layers"], compute_dtype="bf16").to(dev).train()
ts = sat.TrainStep(model)
enc = model.encoder
depth = enc.lookahead_depth
nb = depth + 1
images, caps, lengths = bench.synth_batch(torch, CFG["batch"], CFG["vocab"], CFG["cap_len"], CFG["image"], dev, 123)
batches = [images] + [bench.synth_batch(torch, CFG["batch"], CFG["vocab"], CFG["cap_len"], CFG["image"], dev, 977 * (k + 1))[0] for k in range(nb - 1)]
EV = []


def mark(tag, stream=None):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream if stream is not None else torch.cuda.current_stream())
    EV.append((tag, e))


orig_prefetch = enc.prefetch


def prefetch(images):
    if images is None or any(e[0] is images for e in enc._inflight) or len(enc._inflight) >= enc.lookahead_depth:
        return False
    busy = {e[1] for e in enc._inflight}
    inst = next(i for i in range(enc.lookahead_depth) if i not in busy)
    st = M.lookahead_stream(images.device, inst)
    ok = orig_prefetch(images)
    return ok


# wrap ConvStackProgram.run to bracket it with events on whatever stream is current
from importlib import import_module
R = import_module("show-and-tell_amd.resnet")
orig_run = R.ConvStackProgram.run


def run(self, images):
    mark("stack_start")
    out = orig_run(self, images)
    mark("stack_end")
    return out


R.ConvStackProgram.run = run
orig_pooled = ts._encoder_pooled


def pooled(images, out):
    mark("step_start")
    r = orig_pooled(images, out)
    mark("pooled_ready")
    return r


ts._encoder_pooled = pooled
orig_opt = ts.optimizer_step


def opt(lr=None):
    mark("bwd_done")
    orig_opt(lr)
    mark("adam_done")


ts.optimizer_step = opt


def steps(n):
    for i in range(n):
        nxt = [batches[j % nb] for j in range(i + 1, i + 1 + depth) if j < n]
        ts.step(batches[i % nb], caps, lengths, next_images=nxt or None)


steps(6)
torch.cuda.synchronize()
EV.clear()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps(n)
torch.cuda.synchronize()
base = EV[0][1]
rows = [(base.elapsed_time(e), tag) for tag, e in EV]
# per step: start, pooled, bwd_done, adam_done
cur = {}
stepno = 0
print("times in ms from the first step's start; main-stream phases per step")
for t, tag in rows:
    if tag == "step_start":
        cur = {"s": t}
    elif tag == "pooled_ready":
        cur["p"] = t
    elif tag == "bwd_done":
        cur["b"] = t
    elif tag == "adam_done":
        cur["a"] = t
        print("step %2d: start %7.3f  wait+copy pooled %6.3f  head+decoder fwd+bwd %6.3f  adam %6.3f  | end %7.3f"
              % (stepno, cur["s"], cur["p"] - cur["s"], cur["b"] - cur["p"], cur["a"] - cur["b"], cur["a"]))
        stepno += 1
st = [t for t, tag in rows if tag == "stack_start"]
en = [t for t, tag in rows if tag == "stack_end"]
print("conv stacks (side streams): start, end, duration")
for a, b in zip(st, en):
    print("  %7.3f -> %7.3f  (%6.3f)" % (a, b, b - a))
print("steady state: %.3f ms/step" % ((rows[-1][0] - rows[0][0]) / n))
