#!/bin/bash
# helper for gpurun: conv autotune with 1 / 2 / 3 copies of each candidate in flight (SAT_TUNE_COPIES), default bench
for c in 1 2 3 1 2 3; do
  SAT_TUNE_COPIES=$c python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('copies=$c', d['value'], d['ms_per_step'], 'seq', d['sequential_schedule']['value'], 'frac', d['roofline']['frac'])"
done
