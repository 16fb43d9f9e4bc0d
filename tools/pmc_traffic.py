#!/usr/bin/env python3
"""HBM traffic of one training step from two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>  -> JSON on stdout
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide coalesced reads ->
x2; WRITE_SIZE is exact.  Both counters are in KB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_breakdown import short  # noqa: E402


def last_step(d, counter):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    disp = {}
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != counter:
            continue
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "v": 0.0})
        e["v"] += float(r["Counter_Value"])
    rows = sorted(disp.values(), key=lambda e: e["start"])
    ip = [i for i, e in enumerate(rows) if "image_prep" in e["name"]]
    adam = [i for i, e in enumerate(rows) if "clamp_adam" in e["name"]]
    last = adam[-1]            # last FULL training step (the roofline passes after it run the encoder without an optimizer step)
    s = [i for i in ip if i < last][-1]
    en = [last]
    agg = defaultdict(lambda: [0, 0.0])
    for e in rows[s:en[0] + 1]:
        k = short(e["name"])
        k = "conv_glds_kernel" if k.startswith("conv<") else k
        agg[k][0] += 1
        agg[k][1] += e["v"]
    return agg


def main():
    f, w = last_step(sys.argv[1], "FETCH_SIZE"), last_step(sys.argv[2], "WRITE_SIZE")
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "show-and-tell_amd", "libsat_hip.so")
    out = {"libsat_hip_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],      # bench.py quotes the file only for this build
           "_note": "one training step of bench.py (cfg2); bytes = FETCH_SIZE_KB x 1024 x 2 (gfx950 correction) + WRITE_SIZE_KB x 1024",
           "kernels": {}}
    tot = 0.0
    for k in sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, [0, 0.0])[1])):
        b = (f[k][1] * 2 + w.get(k, [0, 0.0])[1]) * 1024.0
        tot += b
        out["kernels"][k] = {"launches": f[k][0], "FETCH_SIZE_KB_raw": round(f[k][1], 1), "WRITE_SIZE_KB": round(w.get(k, [0, 0.0])[1], 1),
                             "hbm_bytes_corrected": round(b), "hbm_bytes_per_launch_corrected": round(b / max(1, f[k][0]))}
    out["whole_step_hbm_bytes_corrected"] = round(tot)
    c = out["kernels"].get("conv_glds_kernel")
    if c:
        out["kernel"] = "conv_glds_kernel + conv_xp_kernel + conv_pr_kernel (all conv launches), one training step"
        out["launches"] = c["launches"]
        out["hbm_bytes_per_step_corrected"] = c["hbm_bytes_corrected"]
        out["hbm_bytes_per_launch_corrected"] = c["hbm_bytes_per_launch_corrected"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
