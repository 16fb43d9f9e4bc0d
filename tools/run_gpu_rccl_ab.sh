#!/bin/bash
# helper for gpurun: what the data-parallel wrapper costs on ONE rank (--force-dist: RCCL initialised, the bucketed async
# all-reduces issued with world size 1 -- RCCL launches no kernel then, so this isolates the host / stream / queue side)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export SAT_TUNE_FILE=/tmp/tune_base.json
rm -f $SAT_TUNE_FILE
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])'
A="--no-cpu-baseline --no-f32-mode --repeats 3"
python bench.py $A 2>/dev/null | python -c "$P" "no RCCL                        "
python bench.py $A --force-dist 2>/dev/null | python -c "$P" "force-dist                     "
SAT_FORCE_DIST_INIT_ONLY=1 python bench.py $A --force-dist 2>/dev/null | python -c "$P" "RCCL initialised, no all-reduce "
GPU_MAX_HW_QUEUES=16 python bench.py $A --force-dist 2>/dev/null | python -c "$P" "force-dist, 16 hw queues       "
GPU_MAX_HW_QUEUES=16 SAT_FORCE_DIST_INIT_ONLY=1 python bench.py $A --force-dist 2>/dev/null | python -c "$P" "init only, 16 hw queues        "
SAT_LOOKAHEAD_DEPTH=2 python bench.py $A --force-dist 2>/dev/null | python -c "$P" "force-dist, depth 2            "
