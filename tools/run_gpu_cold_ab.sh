#!/bin/bash
# A/B on one box: the conv autotuner times warm back-to-back replays (0) or launches whose input was just rewritten (1)
for rep in 1 2; do
for c in 0 1; do
  for la in "" "--no-lookahead"; do
    echo "== SAT_TUNE_COLD=$c $la"
    SAT_TUNE_COLD=$c timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $la 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f  conv-seq ms %.3f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel'], d['roofline']['frac']))" || exit 1
  done
done
done
