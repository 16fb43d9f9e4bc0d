#!/bin/bash
# helper for gpurun: bench line + rocprofv3 kernel trace of the same command
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python bench.py --steps ${STEPS:-20} --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?
cat gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mode > $R/gpurun_out/prof_run.log 2>&1
rc=$?
tail -n 3 $R/gpurun_out/prof_run.log
find $R/gpurun_out/prof -name "*stats*" | head
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -n 25 "$f"
# keep the big trace out of the merge-back budget
find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
exit $rc
