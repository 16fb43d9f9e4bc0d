#!/bin/bash
# one rank WITH the RCCL bucket all-reduces (--force-dist: the N > 1 code path on one GPU) at look-ahead depth 3 / 2, against the plain N = 1 path
for rep in 1 2; do
for cfg in "3 0" "3 1" "2 1"; do
  set -- $cfg
  fd=""; [ "$2" = "1" ] && fd="--force-dist"
  echo "== SAT_LOOKAHEAD_DEPTH=$1 $fd"
  SAT_LOOKAHEAD_DEPTH=$1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $fd 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || exit 1
done
done
