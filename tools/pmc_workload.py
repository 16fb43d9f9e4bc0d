"""The workload of the counter passes (tools/run_gpu_pmc.sh, tools/run_gpu_pmc_mfma.sh), eager launches so that every kernel is
its own dispatch, cut into segments by marker launches (`kept_tokens`):
    marker | ONE in-order pass of the GROUPED look-ahead program (G batches per launch: what bench.py's timed steps run) | marker |
    ONE whole training step, strictly sequential (ungrouped program + head + decoder + backward + clamp/Adam) | marker
tools/pmc_traffic.py / tools/pmc_summary.py read the segments."""
import importlib
import os
import sys

os.environ["SAT_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
lib = L.load()
wl = bench.WORKLOADS[os.environ.get("SAT_PMC_WORKLOAD", "train")]          # "inception": BASELINE configs[3]
dev = torch.device("cuda", 0)
torch.manual_seed(123)
model = sat.ShowAndTell(wl["embed"], wl["hidden"], wl["vocab"], wl["layers"], compute_dtype="bf16",
                        **({"arch": wl["arch"]} if wl["arch"] else {})).to(dev).train()
ts = sat.TrainStep(model, lr=1e-3, grad_clip=0.1)
images, caps, lengths = bench.synth_batch(torch, wl["batch"], wl["vocab"], wl["cap_len"], wl["image"], dev, 123)
G = model.encoder.lookahead_groups
prog = model.encoder._program(images, instance="g_pmc", groups=G)
arg = [images] + [images.clone() for _ in range(G - 1)]
ids = torch.zeros(4, 4, dtype=torch.int64, device=dev)
kept = torch.zeros(4, dtype=torch.int32, device=dev)


def marker():
    """a launch nothing else in this script makes (the training step has its own `validate_ids`)"""
    L.check(lib.sat_kept_tokens(ids.data_ptr(), ids.stride(0), 4, 4, 2, kept.data_ptr(), L.stream()))


for _ in range(2):                      # warm: both statistics parities, autotune done
    prog.run(arg if G > 1 else images)
    ts.step(images, caps, lengths)
torch.cuda.synchronize()
marker()
prog.run(arg if G > 1 else images)
marker()
ts.step(images, caps, lengths)
marker()
torch.cuda.synchronize()
ts.check_ids()
print("pmc workload done: grouped program of %d batches, then one sequential training step" % G)
