#!/bin/bash
# N = 1 path: look-ahead depth x side streams (fewer streams than stacks: later stacks queue behind earlier ones)
for rep in 1 2; do
for cfg in "3 3" "4 3" "5 3" "4 4"; do
  set -- $cfg
  echo "== SAT_LOOKAHEAD_DEPTH=$1 SAT_LOOKAHEAD_STREAMS=$2"
  SAT_LOOKAHEAD_DEPTH=$1 SAT_LOOKAHEAD_STREAMS=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || exit 1
done
done
