#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export SAT_TUNE_FILE=/tmp/tune_base.json
rm -f $SAT_TUNE_FILE
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])'
A="--no-cpu-baseline --no-f32-mode --repeats 3"
python bench.py $A 2>/dev/null | python -c "$P" "(table)            "
for round in 1 2; do
  python bench.py $A 2>/dev/null | python -c "$P" "default            "
  SAT_MAIN_STREAM_PRIO=-1 python bench.py $A 2>/dev/null | python -c "$P" "main stream high   "
  SAT_MAIN_STREAM_PRIO=0 python bench.py $A 2>/dev/null | python -c "$P" "main stream new, 0 "
  GPU_MAX_HW_QUEUES=4 python bench.py $A 2>/dev/null | python -c "$P" "4 hw queues        "
  GPU_MAX_HW_QUEUES=16 python bench.py $A 2>/dev/null | python -c "$P" "16 hw queues       "
done
