#!/usr/bin/env python3
"""Per-kernel-class counter sums of one segment of tools/pmc_workload.py under `rocprofv3 --pmc`: "program" = one in-order pass of
the grouped look-ahead program (default), "step" = one whole sequential training step.
    python tools/pmc_summary.py <rocprof output dir> [program|step]   -> JSON on stdout"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_breakdown import short  # noqa: E402


def main():
    d = sys.argv[1]
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    disp = {}
    for r in csv.DictReader(open(cc[0])):
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]),
                                               "end": int(r["End_Timestamp"]), "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = sorted(disp.values(), key=lambda e: e["start"])
    # tools/pmc_workload.py: marker | one pass of the grouped look-ahead program | marker | one sequential training step | marker
    marks = [i for i, e in enumerate(rows) if "kept_tokens" in e["name"]][-3:]
    seg = sys.argv[2] if len(sys.argv) > 2 else "program"
    a, b = (marks[0], marks[1]) if seg == "program" else (marks[1], marks[2])
    step = [e for e in rows[a + 1:b] if "kept_tokens" not in e["name"]]
    agg = defaultdict(lambda: defaultdict(float))
    for e in step:
        k = short(e["name"])
        k = "conv (all kernels)" if k.startswith("conv<") else k
        agg[k]["launches"] += 1
        agg[k]["duration_us"] += (e["end"] - e["start"]) / 1e3
        for n, v in e["c"].items():
            agg[k][n] += v
    out = {"_segment": seg, "_note": "tools/pmc_workload.py (cfg2) under rocprofv3 --pmc; SQ_VALU_MFMA_BUSY_CYCLES sums the matrix "
                    "pipes' busy cycles over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16, 64 per "
                    "v_mfma_f32_32x32x2_f32); mfma_busy_frac prices every SIMD for the kernels' whole duration at the "
                    "2.4 GHz peak clock (the chip holds 1.5-2.1 GHz under MFMA load, so true occupancy is higher)"}
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["duration_us"]):
        e = {n: (round(x, 1) if n == "duration_us" else int(x)) for n, x in v.items()}
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            e["mfma_busy_frac"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["duration_us"] * 1e3 * 2.4 * 1024.0), 4)
        out[k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
