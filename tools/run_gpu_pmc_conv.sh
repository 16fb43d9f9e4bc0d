#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/microbench.py conv
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pc$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pc$i -o pc -- python3 $R/tools/microbench.py conv > $R/gpurun_out/pc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pc$i.log; }
  find $R/gpurun_out/pc$i -name "*kernel_trace.csv" -delete
done
ls $R/gpurun_out/pc*/ | head
