"""Print the conv autotuner's per-variant timings (SAT_TUNE_VERBOSE) for the ResNet-152 program at batch 64."""
import importlib, os, sys
os.environ["SAT_TUNE_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sat = importlib.import_module("show-and-tell_amd")
torch.manual_seed(1)
m = sat.ShowAndTell(256, 512, 10000, 1, compute_dtype="bf16").cuda().train()
x = torch.randn(64, 3, 224, 224, device="cuda")
m.encoder._pooled_raw(x)
torch.cuda.synchronize()
