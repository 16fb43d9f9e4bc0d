#!/bin/bash
# A/B on one box: bn3 + residual add + ReLU as its own launch (0) or formed inside the next block's conv1 by conv_du_kernel where it can run (SAT_FUSE_RESIDUAL=2)
for rep in 1 2; do
for f in 0 2; do
  for la in "" "--no-lookahead"; do
    echo "== SAT_FUSE_RESIDUAL=$f $la"
    SAT_FUSE_RESIDUAL=$f timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --repeats 3 $la 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  img/s %.0f  ms/step %.3f  conv-seq ms %.3f  loss %.4f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel'], d['config']['final_loss']))" || exit 1
  done
done
done
