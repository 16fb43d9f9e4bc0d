#!/bin/bash
# look-ahead depth and BatchNorm-apply grid cap after the round-3 conv kernels (one box, interleaved)
for rep in 1 2; do
for cfg in "3 768" "2 768" "4 768" "3 512" "3 1024"; do
  set -- $cfg
  echo "== SAT_LOOKAHEAD_DEPTH=$1 SAT_EW_GRID_CAP=$2"
  SAT_LOOKAHEAD_DEPTH=$1 SAT_EW_GRID_CAP=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || exit 1
done
done
