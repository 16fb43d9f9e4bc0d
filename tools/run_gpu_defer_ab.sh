#!/bin/bash
# helper for gpurun: bn3 + add + ReLU deferred into the next conv1 (conv_ay_kernel, SAT_DEFER_BN3) -- tests, a variant table that
# knows both program forms, then interleaved A/B of the headline step on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_conv_ay.py -m gpu -q -x > gpurun_out/t_ay.log 2>&1; tail -3 gpurun_out/t_ay.log
SAT_DEFER_BN3=0 python tools/dump_tune_table.py gpurun_out/tune_defer0.json 2> gpurun_out/dump0.err || { tail -5 gpurun_out/dump0.err; exit 1; }
python tools/dump_tune_table.py gpurun_out/tune_defer1.json 2> gpurun_out/dump1.err || { tail -5 gpurun_out/dump1.err; exit 1; }
python - <<'PY'
import json
a = json.load(open("gpurun_out/tune_defer0.json")); b = json.load(open("gpurun_out/tune_defer1.json"))
t = dict(a["table"]); t.update(b["table"]); b["table"] = {k: t[k] for k in sorted(t)}
json.dump(b, open("gpurun_out/gfx950.json", "w"), indent=1); open("gpurun_out/gfx950.json", "a").write("\n")
print(len(a["table"]), len(b["table"]), "geometries")
PY
cp gpurun_out/gfx950.json show-and-tell_amd/tune/gfx950.json
for rep in 1 2; do
  for cfg in "0 0" "1 0" "1 1"; do
    set -- $cfg
    SAT_DEFER_BN3=$1 SAT_DEFER_INPLACE=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode > gpurun_out/defer_ab_$1_$2_$rep.json 2> gpurun_out/defer_ab_$1_$2_$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/defer_ab_$1_$2_$rep.json")); r = d["roofline"]
print("defer $1 inplace $2 rep $rep: %.0f img/s %.3f ms/step (min %.0f max %.0f) | conv launches %.3f ms per pass, frac %.4f" % (
    d["value"], d["ms_per_step"], d["repeats"]["images_per_sec_min"], d["repeats"]["images_per_sec_max"], r["ms_per_program_pass_in_kernel"], r["frac"]))
PY
  done
done
