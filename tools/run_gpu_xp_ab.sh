#!/bin/bash
# A/B on one box: the round-3 conv kernels offered to the tuner or not (SAT_CONV_NO_XP / SAT_CONV_NO_PR); look-ahead and sequential steps
for rep in 1 2; do
for cfg in "0 0" "0 1" "1 1"; do
  set -- $cfg
  for la in "" "--no-lookahead"; do
    echo "== SAT_CONV_NO_XP=$1 SAT_CONV_NO_PR=$2 $la"
    SAT_CONV_NO_XP=$1 SAT_CONV_NO_PR=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $la 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f  conv-seq ms %.3f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel'], d['roofline']['frac']))" || exit 1
  done
done
done
