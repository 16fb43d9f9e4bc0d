#!/bin/bash
# the N > 1 code path on one rank (bench.py --force-dist: RCCL initialised, three bucket all-reduces per step): look-ahead depth x side streams
for rep in 1 2; do
for cfg in "2 2" "3 2" "4 2" "3 3"; do
  set -- $cfg
  echo "== SAT_LOOKAHEAD_DEPTH=$1 SAT_LOOKAHEAD_STREAMS=$2 --force-dist"
  SAT_LOOKAHEAD_DEPTH=$1 SAT_LOOKAHEAD_STREAMS=$2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 --force-dist 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || exit 1
done
done
