#!/bin/bash
# helper for gpurun: the round's final records on ONE box -- counter passes first (so that bench.py finds `roofline.traffic` of this
# very library build), then the contract lines of the three workloads and the traces
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
bash tools/run_gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1 && cp gpurun_out/pmc_traffic.json profiles/r05_pmc_traffic.json && cp gpurun_out/pmc_traffic.json gpurun_out/r05_pmc_traffic.json
tail -c 400 gpurun_out/pmc_run.log
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
SAT_PMC_WORKLOAD=inception bash tools/run_gpu_pmc.sh > gpurun_out/pmc_run_cfg3.log 2>&1 && cp gpurun_out/pmc_traffic.json profiles/r05_cfg3_pmc_traffic.json && cp gpurun_out/pmc_traffic.json gpurun_out/r05_cfg3_pmc_traffic.json
tail -c 300 gpurun_out/pmc_run_cfg3.log
bash tools/run_gpu_pmc_mfma.sh > gpurun_out/pmc_mfma_run.log 2>&1; tail -c 300 gpurun_out/pmc_mfma_run.log
STEPS=20 bash tools/run_gpu_bench.sh > gpurun_out/bench_run.log 2>&1; tail -c 600 gpurun_out/bench_run.log
cd $R
timeout -k 10 500 python bench.py --workload inception > gpurun_out/bench_inception.json 2> gpurun_out/bench_inception.err
timeout -k 10 500 python bench.py --workload decode > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err
timeout -k 10 500 python bench.py --workload decode --beam 1 > gpurun_out/bench_decode_greedy.json 2> gpurun_out/bench_decode_greedy.err
for f in bench_inception bench_decode bench_decode_greedy; do python - <<PY
import json
d = json.load(open("gpurun_out/$f.json")); r = d["roofline"]
print("$f: %.0f %s, %.3f ms/step, conv frac %.4f, traffic %s" % (d["value"], d["unit"], d["ms_per_step"], r["frac"], r.get("traffic")))
PY
done
