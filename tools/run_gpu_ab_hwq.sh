#!/bin/bash
# helper for gpurun: does the number of HIP hardware queues matter once RCCL's stream joins the two look-ahead streams and the main one?
for q in 4 8 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --force-dist --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq=$q force-dist', d['value'], d['ms_per_step'])"
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq=$q plain', d['value'], d['ms_per_step'])"
done
