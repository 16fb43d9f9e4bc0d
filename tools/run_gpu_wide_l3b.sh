#!/bin/bash
# helper for gpurun: wide tiles on layer-3 conv3 (K=256 -> N=1024) as well, under look-ahead
export SAT_TUNE_FILE=/tmp/tune_l3.json
rm -f $SAT_TUNE_FILE
python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tuned look-ahead', d['value'], d['ms_per_step'], d['roofline']['frac'])"
for combo in "29 0" "29 28" "29 29" "29 39" "0 39" "0 29"; do
set -- $combo
python - <<PY
import json
t = json.load(open("/tmp/tune_l3.json"))
for k in list(t):
    f = k.split(",")
    if f[0] == "64" and f[1] == "14" and f[6] == "256" and f[9] == "1" and $1: t[k] = $1
    if f[0] == "64" and f[1] == "14" and f[3] == "256" and f[6] == "1024" and $2: t[k] = $2
json.dump(t, open("/tmp/tune_l3_wide.json", "w"))
PY
SAT_TUNE_FILE=/tmp/tune_l3_wide.json python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv1/2=$1 conv3=$2 look-ahead', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'])"
done
