"""Per-dispatch counters of ONE conv kernel, in the program vs replayed (tools/conv1_counters.py under rocprofv3 --pmc).
    python tools/conv1_counters_summary.py <dir with one sub-directory per counter pass> -> JSON on stdout
Dispatches are split at the marker launch (`validate_ids`): before it the whole encoder program ran in order, after it the one op
was replayed back to back.  The kernel is identified as the one the replay launches; in the program only its launches with the
replay's grid AND a BatchNorm-apply launch right in front of them count (layer-3 conv1 behind the previous bottleneck's
normalise+add)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def one_pass(d):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        return None
    disp = {}
    for r in csv.DictReader(open(cc[0])):
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]),
                                               "grid": int(r["Grid_Size"]), "c": defaultdict(float)})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = sorted(disp.values(), key=lambda e: e["start"])
    mark = max(i for i, e in enumerate(rows) if "validate_ids" in e["name"])
    replay = rows[mark + 1:]
    name, grid = replay[-1]["name"], replay[-1]["grid"]
    rep = [e for e in replay if e["name"] == name and e["grid"] == grid]
    # the LAST eager pass of the program in front of the marker (run_timed): from its image_prep on
    ip = max(i for i, e in enumerate(rows[:mark]) if "image_prep" in e["name"])
    prog = [e for i, e in enumerate(rows[ip:mark], ip) if e["name"] == name and e["grid"] == grid and "bn_act" in rows[i - 1]["name"]]
    out = {}
    for tag, es in (("in_program", prog), ("replayed", rep[len(rep) // 2:])):
        agg = defaultdict(float)
        for e in es:
            for k, v in e["c"].items():
                agg[k] += v
        out[tag] = {"launches": len(es), "mean_us_under_profiler": round(sum(e["end"] - e["start"] for e in es) / len(es) / 1e3, 2)}
        out[tag].update({k: round(v / len(es), 1) for k, v in agg.items()})
    out["kernel"] = name[:160]
    return out


def main():
    root = sys.argv[1]
    res = {"_note": "per-launch means; counters summed over XCDs/SEs as rocprofv3 reports them; FETCH_SIZE / TCC_EA0_RDREQ need the gfx950 x2 "
                    "correction for wide reads (MI355X_MICROARCH.md)", "passes": {}}
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        if os.path.isdir(d):
            r = one_pass(d)
            if r:
                res["passes"][os.path.basename(d)] = r
    # derived: L2 hit rate per setting
    for p in res["passes"].values():
        for tag in ("in_program", "replayed"):
            e = p[tag]
            if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
                e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
