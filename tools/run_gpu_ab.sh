#!/bin/bash
# helper for gpurun: quick A/B of conv kernel variants through bench.py (no CPU baseline)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/ab.err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('img/s',d['value'],'ms/step',d['ms_per_step'],'conv_ms',d['roofline']['ms_per_step_in_kernel'],'TF',d['roofline']['achieved'],'loss',d['config']['final_loss'])"
  rc=$?; if [ $rc -ne 0 ]; then tail -5 gpurun_out/ab.err; fi
done
