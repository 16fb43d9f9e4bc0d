"""Layer-3 conv1 of ResNet-152 (M 12544, N 256, K 1024; 35 launches per step) IN THE PROGRAM (its operand was just written by the
previous bottleneck's normalise+add launch) against the same op REPLAYED back to back (operand read a moment ago by the same
launch).  Run under `rocprofv3 --kernel-trace --pmc <counters>` (tools/run_gpu_conv1_counters.sh); tools/conv1_counters_summary.py
splits the dispatches of that kernel at the marker launch.  Without a profiler it prints both timings.
    python tools/conv1_counters.py [Cin Cout KH]      (default 1024 256 1)"""
import ctypes as C
import importlib
import os
import sys

os.environ["SAT_GRAPH"] = "0"           # eager launches: every kernel of the program is its own dispatch in the trace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
lib = L.load()
cin, cout, kh = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (1024, 256, 1)
torch.manual_seed(1)
m = sat.ShowAndTell(256, 512, 10000, 1, compute_dtype="bf16").cuda().train()
x = torch.randn(64, 3, 224, 224, device="cuda")
enc = m.encoder
prog = enc._program(x)
for _ in range(3):
    prog.run(x)
torch.cuda.synchronize()
_, us = prog.run_timed(x)
idx = [i for i in range(prog.n_ops) if prog.ops[i].kind == L.OP_CONV]
sel = [k for k, i in enumerate(idx) if prog.ops[i].Cin == cin and prog.ops[i].Cout == cout and prog.ops[i].KH == kh
       and prog.ops[i].Hout == 14 and prog.ops[i].stride == 1]
in_prog = [us[k] for k in sel]
print("in the program: %d launches, mean %.2f us (min %.2f max %.2f), variant %d" %
      (len(sel), sum(in_prog) / len(in_prog), min(in_prog), max(in_prog), prog.ops[idx[sel[1]]].variant), flush=True)
# marker: one launch no other part of this script makes
ids = torch.zeros(4, 4, dtype=torch.int64, device="cuda")
status = torch.zeros(1, dtype=torch.int32, device="cuda")
L.check(lib.sat_validate_ids(ids.data_ptr(), ids.stride(0), 4, 4, 0, 10, status.data_ptr(), L.stream()))
torch.cuda.synchronize()
op = prog.ops[idx[sel[1]]]
reps = 30
rep = (L.SatOp * reps)(*[op for _ in range(reps)])
L.check(lib.sat_run_ops(rep, reps, L.stream()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.check(lib.sat_run_ops(rep, reps, L.stream()))
e1.record()
torch.cuda.synchronize()
print("replayed back to back: %.2f us" % (e0.elapsed_time(e1) / reps * 1e3), flush=True)
