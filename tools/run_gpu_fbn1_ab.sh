#!/bin/bash
# A/B on one box: bn1 + ReLU as its own launch (0) or applied to conv_pr_kernel's LDS-resident patch where that kernel can run (2)
for rep in 1 2; do
for f in 0 2; do
  for la in "" "--no-lookahead"; do
    echo "== SAT_FUSE_BN1=$f $la"
    SAT_FUSE_BN1=$f timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $la 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f  conv-seq ms %.3f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel'], d['roofline']['frac']))" || exit 1
  done
done
done
