#!/bin/bash
# A/B on one box: the conv autotuner times each candidate alone (1) or as 2 / 3 copies in flight on as many streams (SAT_TUNE_COPIES)
for rep in 1 2; do
for c in 1 2 3; do
  echo "== SAT_TUNE_COPIES=$c"
  SAT_TUNE_COPIES=$c timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f  conv-seq ms %.3f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel'], d['roofline']['frac']))" || exit 1
done
done
