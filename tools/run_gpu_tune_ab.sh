#!/bin/bash
# helper for gpurun: bench.py with the timing tuner on for geometries the committed table does not name (new kernel variants), the
# tuner's in-program choices, and the same line from a second run that loads the first run's table
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
export SAT_AUTOTUNE=${SAT_AUTOTUNE:-force} SAT_TUNE_FILE=$R/gpurun_out/tune_ab.json SAT_TUNE_VERBOSE=1
rm -f $SAT_TUNE_FILE
for rep in 1 2; do
  timeout -k 10 500 python bench.py --no-cpu-baseline --no-f32-mode ${BENCH_ARGS} > gpurun_out/tune_ab_${rep}.json 2> gpurun_out/tune_ab_${rep}.err || { tail -5 gpurun_out/tune_ab_${rep}.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/tune_ab_${rep}.json"))
r = d["roofline"]
print("rep ${rep}: %.0f img/s %.3f ms/step (min %.0f max %.0f) | conv launches %.3f ms per pass, frac %.4f" % (d["value"], d["ms_per_step"], d["repeats"]["images_per_sec_min"], d["repeats"]["images_per_sec_max"], r["ms_per_program_pass_in_kernel"], r["frac"]))
PY
done
grep "tune in program" gpurun_out/tune_ab_1.err | grep ",g2:" | head -40
