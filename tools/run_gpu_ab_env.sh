#!/bin/bash
# helper for gpurun: runtime knobs A/B on the default bench (img/s, ms/step)
run() { env "$@" python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])"; }
run A=0
run HIP_FORCE_DEV_KERNARG=1
run GPU_MAX_HW_QUEUES=16
run SAT_GRAPH=0
run A=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
