#!/bin/bash
# helper for gpurun: is the look-ahead step bound by memory traffic?  Timing-only probes (results are garbage): drop the
# normalise+add+ReLU launches, and additionally the raw conv3 stores -- the upper bound of any fusion that removes them.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export SAT_TUNE_FILE=/tmp/tune_base.json
rm -f $SAT_TUNE_FILE
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "seq", d.get("sequential_schedule",{}).get("ms_per_step"))'
A="--no-cpu-baseline --repeats 3"
python bench.py $A 2>/dev/null | python -c "$P" "default                   "
python bench.py $A 2>/dev/null | python -c "$P" "default                   "
SAT_DBG_SKIP_BN_ADD=1 python bench.py $A 2>/dev/null | python -c "$P" "no bn_add                 "
SAT_DBG_SKIP_BN_ADD=1 SAT_CONV_DBG=8 python bench.py $A 2>/dev/null | python -c "$P" "no bn_add, no conv3 store "
SAT_CONV_DBG=8 python bench.py $A 2>/dev/null | python -c "$P" "no conv3 store            "
