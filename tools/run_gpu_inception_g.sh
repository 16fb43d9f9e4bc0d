#!/bin/bash
# helper for gpurun: BASELINE configs[3] with more batches per grouped launch (the geometries of G = 3 / 4 are not in the committed
# table: timed on the fly, SAT_AUTOTUNE=1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
export SAT_AUTOTUNE=1 SAT_TUNE_FILE=/tmp/inc_tune.json
for cfg in "2 4" "3 6" "4 4" "4 8"; do
  set -- $cfg
  SAT_LOOKAHEAD_GROUPS=$1 SAT_LOOKAHEAD_DEPTH=$2 timeout -k 10 500 python bench.py --workload inception --no-cpu-baseline --no-f32-mode > gpurun_out/inc_g_$1_$2.json 2> gpurun_out/inc_g_$1_$2.err || { tail -5 gpurun_out/inc_g_$1_$2.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/inc_g_$1_$2.json"))
r = d["roofline"]
print("groups $1 depth $2: %.0f img/s %.3f ms/step (min %.0f max %.0f) | conv launches %.3f ms per pass, frac %.4f, batches/launch %s" % (d["value"], d["ms_per_step"], d["repeats"]["images_per_sec_min"], d["repeats"]["images_per_sec_max"], r["ms_per_program_pass_in_kernel"], r["frac"], r["batches_per_launch"]))
PY
done
