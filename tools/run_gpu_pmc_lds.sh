#!/bin/bash
# helper for gpurun: LDS / wait-state counters of the bench (own --pmc passes, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_lds$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_lds$i -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-mode --repeats 1 > $R/gpurun_out/pmc_lds$i.log 2>&1
  rc=$?; tail -n 1 $R/gpurun_out/pmc_lds$i.log | cut -c1-160; if [ $rc -ne 0 ]; then echo "pass $i failed rc=$rc"; exit $rc; fi
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_lds$i > $R/gpurun_out/pmc_lds${i}_summary.json
  find $R/gpurun_out/pmc_lds$i -name "*kernel_trace.csv" -delete
  find $R/gpurun_out/pmc_lds$i -name "*counter_collection.csv" -size +20M -delete
done
python3 - <<PY
import json
for i in (1,2):
    d=json.load(open("$R/gpurun_out/pmc_lds%d_summary.json"%i))
    for k in list(d)[1:4]:
        print(k, {a:b for a,b in d[k].items() if a.startswith("SQ_") or a in ("launches","duration_us")})
PY
