#!/bin/bash
# A/B on one box: workgroup size of the table-deriving BatchNorm-apply launches
for rep in 1 2; do
for b in 256 512 1024; do
  for la in "" "--no-lookahead"; do
    echo "== SAT_BN_BLOCK=$b $la"
    SAT_BN_BLOCK=$b timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $la 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || exit 1
  done
done
done
