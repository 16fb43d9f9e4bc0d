#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"])'
A="--no-cpu-baseline --no-f32-mode --repeats 3"
for round in 1 2 3; do
  python bench.py $A 2>/dev/null | python -c "$P" "copies=1 "
  SAT_TUNE_COPIES=2 python bench.py $A 2>/dev/null | python -c "$P" "copies=2 "
done
