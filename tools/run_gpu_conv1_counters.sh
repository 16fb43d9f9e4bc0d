#!/bin/bash
# layer-3 conv1 in the program vs replayed: timings without a profiler, then one rocprofv3 --pmc pass per counter group
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out/c1; cd /tmp; export TMPDIR=/tmp
export SAT_TUNE_FILE=$R/gpurun_out/c1/tune.json
python3 $R/tools/conv1_counters.py > $R/gpurun_out/c1/timing.txt 2>&1 || { tail -5 $R/gpurun_out/c1/timing.txt; exit 1; }
cat $R/gpurun_out/c1/timing.txt
i=0
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1)); rm -rf $R/gpurun_out/c1/p$i
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/c1/p$i -o pc -- python3 $R/tools/conv1_counters.py > $R/gpurun_out/c1/p$i.log 2>&1 || { echo "pass $i ($c) failed"; tail -3 $R/gpurun_out/c1/p$i.log; rm -rf $R/gpurun_out/c1/p$i; }
  find $R/gpurun_out/c1/p$i -name "*kernel_trace.csv" -delete 2>/dev/null
done
python3 $R/tools/conv1_counters_summary.py $R/gpurun_out/c1 > $R/gpurun_out/c1/summary.json
find $R/gpurun_out/c1 -name "*counter_collection.csv" -size +8M -delete
head -c 3000 $R/gpurun_out/c1/summary.json
