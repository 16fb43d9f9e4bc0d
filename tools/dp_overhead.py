#!/usr/bin/env python3
"""Where the one-rank data-parallel wrapper loses time (bench.py --force-dist is ~9 % slower than the plain step although RCCL
launches no kernel at world size 1): the same loop with the real c10d all-reduce, with a no-op stand-in, and with the waits moved."""
import importlib, os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sat = importlib.import_module("show-and-tell_amd")
CFG = bench.CFG
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
saved = os.dup(1); os.dup2(2, 1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)
torch.manual_seed(123)
model = sat.ShowAndTell(CFG["embed"], CFG["hidden"], CFG["vocab"], CFG["layers"], compute_dtype="bf16").to(dev).train()
ts = sat.TrainStep(model)
dp = sat.DataParallelStep(ts)
depth = model.encoder.lookahead_depth
nb = depth + 1
images, caps, lengths = bench.synth_batch(torch, CFG["batch"], CFG["vocab"], CFG["cap_len"], CFG["image"], dev, 123)
batches = [images] + [bench.synth_batch(torch, CFG["batch"], CFG["vocab"], CFG["cap_len"], CFG["image"], dev, 977 * (k + 1))[0] for k in range(nb - 1)]
gt = sum(l - 1 for l in lengths)


def run(n):
    for i in range(n):
        nxt = [batches[j % nb] for j in range(i + 1, i + 1 + depth) if j < n]
        dp.step((batches[i % nb], caps, lengths), gt, next_images=nxt or None)


def timed(label):
    run(6); ts.drop_lookahead(); torch.cuda.synchronize()
    ts_ = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); ts_.append((time.perf_counter() - t0) / 20 * 1e3)
    print("%-52s %.3f ms/step" % (label, sorted(ts_)[1]))


class _NoWork:
    def wait(self):
        return True


real = dist.all_reduce
dp.world = 1
timed("plain step (world 1, no all-reduce calls)")
dp.world = 2
timed("c10d all_reduce(async_op=True) x3 + wait")
dist.all_reduce = lambda *a, **k: _NoWork()
dp.dist = dist
timed("no-op stand-in for all_reduce")
dist.all_reduce = lambda t, **k: (real(t, op=k["op"], group=k["group"], async_op=False), _NoWork())[1]
timed("c10d all_reduce(async_op=False)")
dist.all_reduce = real
dist.destroy_process_group()
