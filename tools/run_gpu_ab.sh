#!/bin/bash
# helper for gpurun: A/B one environment variable on the same box.  usage: run_gpu_ab.sh VAR v1 v2 ...
var=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  for rep in 1 2; do
    r=$(env $var=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$var=$v run$rep: $r"
  done
done
