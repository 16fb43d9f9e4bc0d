#!/bin/bash
# helper for gpurun: throughput-oriented tiles (128x256: fewer, larger workgroups) x look-ahead depth.  Hypothesis: with small grids
# per launch, more stacks in flight raise the chip's throughput (CU-time per launch counts, not one launch's span).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export SAT_TUNE_FILE=/tmp/tune_base.json
rm -f $SAT_TUNE_FILE
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["repeats"])'
python bench.py --no-cpu-baseline --no-f32-mode --repeats 3 2>/dev/null | python -c "$P" "tuned depth3 (makes the table)"
python - <<PY
import json
t = json.load(open("/tmp/tune_base.json"))
n = 0
for k in list(t):
    f = k.split(",")
    if int(f[6]) >= 256:            # Cout >= 256: the 128x256 tile, 3 stages
        t[k] = 29; n += 1
json.dump(t, open("/tmp/tune_wide.json", "w"))
print("forced variant 29 on", n, "of", len(t), "geometries")
PY
for d in 3 4 5 6; do
  SAT_LOOKAHEAD_DEPTH=$d SAT_TUNE_FILE=/tmp/tune_base.json python bench.py --no-cpu-baseline --no-f32-mode --repeats 3 2>/dev/null | python -c "$P" "tuned depth$d"
  SAT_LOOKAHEAD_DEPTH=$d SAT_TUNE_FILE=/tmp/tune_wide.json python bench.py --no-cpu-baseline --no-f32-mode --repeats 3 2>/dev/null | python -c "$P" "wide  depth$d"
done
