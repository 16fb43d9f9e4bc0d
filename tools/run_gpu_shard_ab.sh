#!/bin/bash
# A/B on one box: per-tile statistics slabs + reducer launches for the many-tile convs (0) or 8-shard integer atomics (SAT_SHARDED_BN_MAX_TILES=1600)
for rep in 1 2; do
for f in 0 1600 7000; do
  for la in "" "--no-lookahead"; do
    echo "== SAT_SHARDED_BN_MAX_TILES=$f $la"
    SAT_SHARDED_BN_MAX_TILES=$f timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $la 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f  conv-seq ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel']))" || exit 1
  done
done
done
