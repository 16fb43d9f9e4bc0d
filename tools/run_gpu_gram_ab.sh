#!/bin/bash
# helper for gpurun: the Gram-statistics form of bn3 (SAT_GRAM_BN3, default on) against the three-launch form, bench.py lines
# interleaved on one box, then the kernel trace of sequential steps of the fused form (per-kernel breakdown of one step)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
export SAT_AUTOTUNE=1 SAT_TUNE_FILE=$R/gpurun_out/tune_gram.json
for rep in 1 2; do
  for g in 1 128 0; do
    export SAT_GRAM_MAX_PLANES=512; [ $g = 128 ] && export SAT_GRAM_MAX_PLANES=128
    SAT_GRAM_BN3=$g timeout -k 10 400 python bench.py --no-cpu-baseline --no-f32-mode ${BENCH_ARGS} > gpurun_out/gram_ab_${g}_${rep}.json 2> gpurun_out/gram_ab_${g}_${rep}.err || { tail -5 gpurun_out/gram_ab_${g}_${rep}.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/gram_ab_${g}_${rep}.json"))
r = d["roofline"]
print("SAT_GRAM_BN3=${g} rep ${rep}: %.0f img/s %.3f ms/step (min %.0f max %.0f) | conv launches %.3f ms per pass, frac %.4f" % (d["value"], d["ms_per_step"], d["repeats"]["images_per_sec_min"], d["repeats"]["images_per_sec_max"], r["ms_per_program_pass_in_kernel"], r["frac"]))
PY
  done
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_seq
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_seq -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mode --no-lookahead --repeats 1 > $R/gpurun_out/prof_seq_run.log 2>&1
tail -n 1 $R/gpurun_out/prof_seq_run.log | cut -c1-160
fs=$(find $R/gpurun_out/prof_seq -name "*kernel_trace.csv" | head -1)
[ -n "$fs" ] && python3 $R/tools/step_breakdown.py "$fs" 60 > $R/gpurun_out/one_step_breakdown.txt 2>&1; head -n ${LINES_OUT:-40} $R/gpurun_out/one_step_breakdown.txt
find $R/gpurun_out/prof_seq -name "*kernel_trace.csv" -size +20M -delete
