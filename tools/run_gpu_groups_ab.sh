#!/bin/bash
# grouped look-ahead programs (G batches per launch of the frozen stack) against one program per batch, interleaved on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
out=$R/gpurun_out/groups_ab.txt; : > $out
run() { # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --steps ${STEPS:-20} --no-cpu-baseline --no-f32-mode --repeats 3 > $R/gpurun_out/_ab.json 2> $R/gpurun_out/_ab.err || { echo "$label FAILED" >> $out; tail -5 $R/gpurun_out/_ab.err >> $out; return 1; }
  python3 - "$label" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_ab.json"))
print("%-34s %8.0f img/s  %.3f ms/step  (min %.0f max %.0f)  conv in sequence %.3f ms = %.4f of peak, %d launches, loss %.4f"
      % (sys.argv[1], d["value"], d["ms_per_step"], d["repeats"]["images_per_sec_min"], d["repeats"]["images_per_sec_max"],
         d["roofline"]["ms_per_step_in_kernel"], d["roofline"]["frac"], int(d["roofline"]["kernel"].split("(")[1].split()[0]), d["config"]["final_loss"]))
PY
}
STEPS=${STEPS:-20}
for rep in 1 2; do
  run "groups 1 depth 3 streams 3 (round 3)" SAT_LOOKAHEAD_GROUPS=1 SAT_LOOKAHEAD_DEPTH=3 || exit 1
  run "groups 2 depth 6 streams 3" SAT_LOOKAHEAD_GROUPS=2 SAT_LOOKAHEAD_DEPTH=6 || exit 1
  run "groups 2 depth 6 streams 2" SAT_LOOKAHEAD_GROUPS=2 SAT_LOOKAHEAD_DEPTH=6 SAT_LOOKAHEAD_STREAMS=2 || exit 1
  run "groups 2 depth 6 streams 1" SAT_LOOKAHEAD_GROUPS=2 SAT_LOOKAHEAD_DEPTH=6 SAT_LOOKAHEAD_STREAMS=1 || exit 1
  run "groups 2 depth 4 streams 2" SAT_LOOKAHEAD_GROUPS=2 SAT_LOOKAHEAD_DEPTH=4 || exit 1
  run "groups 2 depth 4 streams 1" SAT_LOOKAHEAD_GROUPS=2 SAT_LOOKAHEAD_DEPTH=4 SAT_LOOKAHEAD_STREAMS=1 || exit 1
  run "groups 4 depth 8 streams 1" SAT_LOOKAHEAD_GROUPS=4 SAT_LOOKAHEAD_DEPTH=8 SAT_LOOKAHEAD_STREAMS=1 || exit 1
  run "groups 4 depth 4 streams 1" SAT_LOOKAHEAD_GROUPS=4 SAT_LOOKAHEAD_DEPTH=4 SAT_LOOKAHEAD_STREAMS=1 || exit 1
done
cat $out
