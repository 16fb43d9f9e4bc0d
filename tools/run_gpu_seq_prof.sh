#!/bin/bash
# helper for gpurun: rocprofv3 kernel trace of the strictly sequential schedule -> per-kernel breakdown of ONE training step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_seq
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_seq -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mode --no-lookahead --repeats 1 > $R/gpurun_out/prof_seq_run.log 2>&1
tail -n 1 $R/gpurun_out/prof_seq_run.log | cut -c1-160
fs=$(find $R/gpurun_out/prof_seq -name "*kernel_trace.csv" | head -1)
[ -n "$fs" ] && python3 $R/tools/step_breakdown.py "$fs" 60 > $R/gpurun_out/one_step_breakdown.txt 2>&1; head -n ${LINES_OUT:-30} $R/gpurun_out/one_step_breakdown.txt
find $R/gpurun_out/prof_seq -name "*kernel_trace.csv" -size +20M -delete
