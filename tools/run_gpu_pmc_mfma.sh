#!/bin/bash
# helper for gpurun: matrix-pipe occupancy counters of the bench (own --pmc pass, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
# the autotuner's timing runs are distorted under the profiler: the profiled passes load the table an un-profiled run saved
export SAT_TUNE_FILE=$R/gpurun_out/pmc_tune.json
[ -f $SAT_TUNE_FILE ] || python3 $R/tools/pmc_workload.py > $R/gpurun_out/pmc_warm.log 2>&1 || { tail -5 $R/gpurun_out/pmc_warm.log; exit 1; }
rm -rf $R/gpurun_out/pmc_mfma
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma -o pmc -- python3 $R/tools/pmc_workload.py > $R/gpurun_out/pmc_mfma.log 2>&1
rc=$?; tail -n 2 $R/gpurun_out/pmc_mfma.log | cut -c1-200; if [ $rc -ne 0 ]; then echo "pmc mfma failed rc=$rc"; exit $rc; fi
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_mfma > $R/gpurun_out/pmc_mfma_summary.json
cat $R/gpurun_out/pmc_mfma_summary.json | head -60
find $R/gpurun_out/pmc_mfma -name "*kernel_trace.csv" -delete
find $R/gpurun_out/pmc_mfma -name "*counter_collection.csv" -size +20M -delete
