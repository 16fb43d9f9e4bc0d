#!/bin/bash
# helper for gpurun: A/B one environment switch on the bench (no CPU baseline, no f32 leg): `bash tools/run_gpu_ab.sh VAR [A B]`
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
V=${1:-SAT_FUSE_RESIDUAL}; A=${2:-0}; B=${3:-1}
for val in $A $B $A $B; do
  env $V=$val timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-mode > gpurun_out/ab_$val.json 2> gpurun_out/ab_$val.err || { echo "bench failed ($V=$val)"; tail -n 20 gpurun_out/ab_$val.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$val.json"))
print("$V=$val", d["value"], "img/s", d["ms_per_step"], "ms/step; conv", d["roofline"]["ms_per_step_in_kernel"], "ms", d["roofline"]["frac"])
PY
done
