#!/bin/bash
# helper for gpurun: look-ahead depth x side streams (more program instances than streams: a later run queues behind an earlier one on
# its stream and starts the moment that one ends), bench.py lines at the contract's 20 steps, interleaved on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
for rep in 1 2; do
  for cfg in ${CFGS:-"6 3" "6 2" "8 3" "8 2" "10 3" "10 2" "12 3"}; do
    set -- $cfg
    SAT_LOOKAHEAD_DEPTH=$1 SAT_LOOKAHEAD_STREAMS=$2 timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-f32-mode > gpurun_out/depth_ab_$1_$2_$rep.json 2> gpurun_out/depth_ab_$1_$2_$rep.err || { tail -5 gpurun_out/depth_ab_$1_$2_$rep.err; continue; }
    python - <<PY
import json
d = json.load(open("gpurun_out/depth_ab_$1_$2_$rep.json"))
print("depth $1 streams $2 rep $rep: %.0f img/s %.3f ms/step (min %.0f max %.0f)" % (d["value"], d["ms_per_step"], d["repeats"]["images_per_sec_min"], d["repeats"]["images_per_sec_max"]))
PY
  done
done
