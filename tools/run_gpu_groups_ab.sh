#!/bin/bash
# grouped look-ahead programs (G batches per launch of the frozen stack) against one program per batch, interleaved on one box
#   CONFIGS="1,3,3 2,6,3 ..." (groups,depth,streams)   STEPS=20   REPS=2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
out=$R/gpurun_out/groups_ab.txt; : > $out
run() { # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --steps ${STEPS:-20} --no-cpu-baseline --no-f32-mode --repeats ${REPEATS:-5} > $R/gpurun_out/_ab.json 2> $R/gpurun_out/_ab.err || { echo "$label FAILED" >> $out; tail -5 $R/gpurun_out/_ab.err >> $out; return 1; }
  python3 - "$label" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_ab.json"))
print("%-34s %8.0f img/s  %.3f ms/step  (min %.0f max %.0f)  conv in sequence %.3f ms/batch = %.4f of peak, loss %.4f"
      % (sys.argv[1], d["value"], d["ms_per_step"], d["repeats"]["images_per_sec_min"], d["repeats"]["images_per_sec_max"],
         d["roofline"]["ms_per_step_in_kernel"], d["roofline"]["frac"], d["config"]["final_loss"]))
PY
}
for rep in $(seq 1 ${REPS:-2}); do
  for c in ${CONFIGS:-1,3,3 2,6,3}; do
    IFS=, read g d s <<< "$c"
    run "groups $g depth $d streams $s" SAT_LOOKAHEAD_GROUPS=$g SAT_LOOKAHEAD_DEPTH=$d SAT_LOOKAHEAD_STREAMS=$s || exit 1
  done
done
cat $out
