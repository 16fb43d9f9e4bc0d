#!/bin/bash
# helper for gpurun: decode tests, then the beam-5 loop alone (wall + rocprofv3 per-kernel averages)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "beam or decode or greedy or sample" > gpurun_out/t_dec.log 2>&1; tail -3 gpurun_out/t_dec.log
python tools/decode_trace.py 5 40 2>/dev/null | tail -1
python tools/decode_trace.py 1 40 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_dec && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -o dec -- python3 $R/tools/decode_trace.py 5 20 > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_dec -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-70s calls %4s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
