#!/bin/bash
# helper for gpurun: force the 128x256 tile (98 workgroups) on the layer-3 conv1 / conv2 launches and compare under look-ahead
export SAT_TUNE_FILE=/tmp/tune_l3.json
rm -f $SAT_TUNE_FILE
python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tuned       look-ahead', d['value'], d['ms_per_step'], d['roofline']['frac'])"
python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tuned(file) look-ahead', d['value'], d['ms_per_step'], d['roofline']['frac'])"
for v in 30 29 28; do
python - <<PY
import json
t = json.load(open("/tmp/tune_l3.json"))
n = 0
for k in list(t):
    f = k.split(",")
    if f[0] == "64" and f[1] == "14" and f[6] == "256" and f[9] == "1":
        t[k] = $v; n += 1
json.dump(t, open("/tmp/tune_l3_wide.json", "w"))
print("forced variant", $v, "on", n, "geometries")
PY
SAT_TUNE_FILE=/tmp/tune_l3_wide.json python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide v$v    look-ahead', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
