#!/bin/bash
# helper for gpurun: A/B of the BatchNorm-apply kernels' cache policy / grid under look-ahead (two interleaved rounds)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export SAT_TUNE_FILE=/tmp/tune_base.json
rm -f $SAT_TUNE_FILE
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "seq", d.get("sequential_schedule",{}).get("ms_per_step"))'
A="--no-cpu-baseline --repeats 3"
python bench.py $A 2>/dev/null | python -c "$P" "(table)                  "
for round in 1 2; do
for cfg in "0 2048" "1 2048" "0 1024" "1 1024" "0 768" "1 768" "0 512" "1 512" "1 384"; do
  set -- $cfg
  SAT_BN_NT=$1 SAT_EW_GRID_CAP=$2 python bench.py $A 2>/dev/null | python -c "$P" "nt=$1 ew_grid_cap=$2 "
done
done
