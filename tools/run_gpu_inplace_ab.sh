#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export SAT_TUNE_FILE=/tmp/tune_base.json
rm -f $SAT_TUNE_FILE
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "seq", d.get("sequential_schedule",{}).get("ms_per_step"))'
A="--no-cpu-baseline --repeats 3"
python bench.py $A 2>/dev/null | python -c "$P" "(table)       "
for round in 1 2; do
  python bench.py $A 2>/dev/null | python -c "$P" "in place      "
  SAT_BN_INPLACE=0 python bench.py $A 2>/dev/null | python -c "$P" "three buffers "
  SAT_LOOKAHEAD_DEPTH=4 python bench.py $A 2>/dev/null | python -c "$P" "in place, depth 4"
done
