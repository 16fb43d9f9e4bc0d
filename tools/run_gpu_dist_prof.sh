#!/bin/bash
# helper for gpurun: kernel trace of the one-rank RCCL path (bench.py --force-dist) to see what the DP wrapper adds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
rm -rf $R/gpurun_out/prof_dist
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dist -o bench -- python3 $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mode --force-dist --repeats 1 > $R/gpurun_out/prof_dist_run.log 2>&1
rc=$?
tail -n 2 $R/gpurun_out/prof_dist_run.log | cut -c1-200
python3 $R/tools/step_breakdown.py $(find $R/gpurun_out/prof_dist -name "*kernel_trace.csv" | head -1) 60 | grep -v "conv<\|bn_\|skinny\|lstm_bwd\|gemm_kernel" 
exit $rc
