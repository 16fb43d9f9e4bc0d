#!/bin/bash
# A/B on one box: three-launch conv3 / two-pass conv3 (SAT_CONV3_TWOPASS=1), conv_xp_kernel offered to the tuner in both
for rep in 1 2; do
for tp in 0 1; do
  for la in "" "--no-lookahead"; do
    echo "== SAT_CONV3_TWOPASS=$tp $la"
    SAT_CONV3_TWOPASS=$tp timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $la 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f  conv-seq ms %.3f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel'], d['roofline']['frac']))" || exit 1
  done
done
done
