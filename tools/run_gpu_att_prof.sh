#!/bin/bash
# helper for gpurun: rocprofv3 kernel trace of the Show-Attend-Tell train step bench (tools/bench_attend.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_att
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_att -o att -- python3 $R/tools/bench_attend.py bf16 > $R/gpurun_out/prof_att.log 2>&1
grep -E "ms/step|captions" $R/gpurun_out/prof_att.log
