#!/bin/bash
# helper for gpurun: HBM traffic counters of the bench (separate --pmc passes, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
# the autotuner's timing runs are distorted under the profiler: the profiled passes load the table an un-profiled run saved
export SAT_TUNE_FILE=$R/gpurun_out/pmc_tune.json
[ -f $SAT_TUNE_FILE ] || python3 $R/tools/pmc_workload.py > $R/gpurun_out/pmc_warm.log 2>&1 || { tail -5 $R/gpurun_out/pmc_warm.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python3 $R/tools/pmc_workload.py > $R/gpurun_out/pmc_$c.log 2>&1
  rc=$?; tail -n 2 $R/gpurun_out/pmc_$c.log; if [ $rc -ne 0 ]; then echo "pmc $c failed rc=$rc"; exit $rc; fi
  ls -la $R/gpurun_out/pmc_$c | head
  find $R/gpurun_out/pmc_$c -name "*kernel_trace.csv" -delete
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE > $R/gpurun_out/pmc_traffic.json
head -c 1500 $R/gpurun_out/pmc_traffic.json
find $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" -size +20M -delete
