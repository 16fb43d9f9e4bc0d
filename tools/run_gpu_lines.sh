#!/bin/bash
# helper for gpurun: the contract lines of the three workloads + the traces, WITHOUT the counter passes (those are bound to the library
# build, not to the box: `profiles/r05_pmc_traffic.json` of the same build is picked up by bench.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
STEPS=20 bash tools/run_gpu_bench.sh > gpurun_out/bench_run.log 2>&1; tail -c 300 gpurun_out/bench_run.log
cd $R
timeout -k 10 500 python bench.py --workload inception > gpurun_out/bench_inception.json 2> gpurun_out/bench_inception.err
timeout -k 10 500 python bench.py --workload decode > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err
timeout -k 10 500 python bench.py --workload decode --beam 1 > gpurun_out/bench_decode_greedy.json 2> gpurun_out/bench_decode_greedy.err
for f in bench bench_inception bench_decode bench_decode_greedy; do python - <<PY
import json
d = json.load(open("gpurun_out/$f.json")); r = d["roofline"]
print("$f: %.0f %s, %.3f ms/step, conv frac %.4f, traffic %s" % (d["value"], d["unit"], d["ms_per_step"], r["frac"], r.get("traffic")))
PY
done
