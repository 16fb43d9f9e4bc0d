#!/bin/bash
# A/B on one box: conv_xp_kernel offered to the tuner (default) or not (SAT_CONV_NO_XP=1); look-ahead and sequential steps
for rep in 1 2; do
for nx in 0 1; do
  for la in "" "--no-lookahead"; do
    echo "== SAT_CONV_NO_XP=$nx $la"
    SAT_CONV_NO_XP=$nx timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $la 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f  conv-seq ms %.3f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel'], d['roofline']['frac']))" || exit 1
  done
done
done
