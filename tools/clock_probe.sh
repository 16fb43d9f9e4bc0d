#!/bin/bash
# helper for gpurun: sample the GPU's clocks and power while bench.py's look-ahead steps run (is the chip holding its clock?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|busy" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clock_probe.txt 2>&1 &
SMI=$!
timeout -k 10 300 python bench.py --steps 200 --warmup 5 --repeats 1 --no-cpu-baseline --no-f32-mode > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err
wait $SMI
python - <<'PY'
import json
d = json.load(open("gpurun_out/clock_bench.json"))
print("200-step region: %.0f img/s %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
cat gpurun_out/clock_probe.txt | cut -c1-400 | sed -n '1,60p'
