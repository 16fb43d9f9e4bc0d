#!/bin/bash
# helper for gpurun: BASELINE configs[3] (Inception-v3 299x299 + 2-layer LSTM hidden 1024) -- grouped look-ahead (2 batches per
# program run, depth 4) against round 4's form (one batch per run, depth 2), bench.py lines interleaved on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
for rep in 1 2; do
  for cfg in "2 4" "1 2" "2 2" "2 6"; do
    set -- $cfg
    SAT_LOOKAHEAD_GROUPS=$1 SAT_LOOKAHEAD_DEPTH=$2 timeout -k 10 500 python bench.py --workload inception --no-cpu-baseline --no-f32-mode > gpurun_out/inc_ab_$1_$2_${rep}.json 2> gpurun_out/inc_ab_$1_$2_${rep}.err || { tail -5 gpurun_out/inc_ab_$1_$2_${rep}.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/inc_ab_$1_$2_${rep}.json"))
r = d["roofline"]
print("groups $1 depth $2 rep ${rep}: %.0f img/s %.3f ms/step (min %.0f max %.0f) | conv launches %.3f ms per pass, frac %.4f, batches/launch %s" % (d["value"], d["ms_per_step"], d["repeats"]["images_per_sec_min"], d["repeats"]["images_per_sec_max"], r["ms_per_program_pass_in_kernel"], r["frac"], r["batches_per_launch"]))
PY
  done
done
