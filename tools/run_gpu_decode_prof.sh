#!/bin/bash
# helper for gpurun: kernel stats of the decode path (tools/bench_decode.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_dec
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -o dec -- python3 $R/tools/bench_decode.py --no-cpu --iters 20 > $R/gpurun_out/prof_dec.log 2>&1
tail -n 1 $R/gpurun_out/prof_dec.log | cut -c1-300
f=$(find $R/gpurun_out/prof_dec -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-90s n=%6s avg %8.1f us tot %8.2f ms" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
find $R/gpurun_out/prof_dec -name "*kernel_trace.csv" -size +20M -delete
