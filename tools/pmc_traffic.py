#!/usr/bin/env python3
"""HBM / fabric traffic from two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_workload.py.
    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>  -> JSON on stdout
Segment 1 (between the first two markers) = one in-order pass of the grouped look-ahead program: the conv launches the timed
steps of bench.py run; segment 2 = one whole sequential training step.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE reports half of the bytes of wide coalesced reads -> x2; WRITE_SIZE is exact.  Both counters are in KB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_breakdown import short  # noqa: E402


def segments(d, counter):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    disp = {}
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != counter:
            continue
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "v": 0.0})
        e["v"] += float(r["Counter_Value"])
    rows = sorted(disp.values(), key=lambda e: e["start"])
    marks = [i for i, e in enumerate(rows) if "kept_tokens" in e["name"]][-3:]
    out = []
    for a, b in ((marks[0], marks[1]), (marks[1], marks[2])):
        agg = defaultdict(lambda: [0, 0.0])
        for e in rows[a + 1:b]:
            if "kept_tokens" in e["name"]:
                continue
            k = short(e["name"])
            k = "conv (all kernels)" if k.startswith("conv<") else k
            if "lstm_persist_bwd" in e["name"]:
                k = "lstm_persist_bwd"
            agg[k][0] += 1
            agg[k][1] += e["v"]
        out.append(agg)
    return out


def table(f, w):
    res, tot = {}, 0.0
    for k in sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, [0, 0.0])[1])):
        b = (f[k][1] * 2 + w.get(k, [0, 0.0])[1]) * 1024.0
        tot += b
        res[k] = {"launches": f[k][0], "FETCH_SIZE_KB_raw": round(f[k][1], 1), "WRITE_SIZE_KB": round(w.get(k, [0, 0.0])[1], 1),
                  "bytes_corrected": round(b), "bytes_per_launch_corrected": round(b / max(1, f[k][0]))}
    return res, round(tot)


def main():
    fs, ws = segments(sys.argv[1], "FETCH_SIZE"), segments(sys.argv[2], "WRITE_SIZE")
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "show-and-tell_amd", "libsat_hip.so")
    out = {"libsat_hip_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],      # bench.py quotes the file only for this build
           "_note": "tools/pmc_workload.py under rocprofv3 --pmc; bytes = FETCH_SIZE_KB x 1024 x 2 (gfx950 correction) + WRITE_SIZE_KB x 1024; "
                    "FETCH_SIZE / WRITE_SIZE count the L2s' memory-side requests: Infinity-Cache hits are included"}
    prog, tot_p = table(fs[0], ws[0])
    step, tot_s = table(fs[1], ws[1])
    out["grouped_program_pass"] = {"kernels": prog, "bytes_corrected": tot_p}
    out["sequential_training_step"] = {"kernels": step, "bytes_corrected": tot_s}
    c = prog.get("conv (all kernels)")
    if c:
        out["kernel"] = "conv_glds_kernel + conv_xp_kernel + conv_pw_kernel + conv_aw_kernel + conv_ap_kernel + conv_pr_kernel + conv_stem_kernel: every conv launch of one pass of the grouped look-ahead program"
        out["launches"] = c["launches"]
        out["hbm_bytes_per_launch_corrected"] = c["bytes_per_launch_corrected"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
