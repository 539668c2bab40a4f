#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + FETCH_SIZE / WRITE_SIZE PMC passes)
into small text/JSON summaries that can be committed under profiles/."""
import csv
import glob
import json
import os
import sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = {}


def find(d, pat):
    return sorted(glob.glob(os.path.join(base, d, "**", pat), recursive=True))


# kernel stats
for f in find("prof_trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = rows[:8]
for f in find("prof_trace", "*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
         if "demod_kernel" in r.get("Kernel_Name", "")]
    if d:
        d_sorted = sorted(d)
        out["demod_kernel_trace"] = {"launches": len(d), "avg_ns": sum(d) / len(d),
                                     "median_ns": d_sorted[len(d) // 2], "min_ns": d_sorted[0],
                                     "max_ns": d_sorted[-1]}
        r0 = [r for r in rows if "demod_kernel" in r.get("Kernel_Name", "")][0]
        out["demod_kernel_resources"] = {k: r0.get(k) for k in
                                         ("VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                          "Workgroup_Size", "Grid_Size", "Scratch_Size")}
# PMC
for name, d in (("FETCH_SIZE", "prof_pmc1"), ("WRITE_SIZE", "prof_pmc2")):
    for f in find(d, "*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        vals = [float(r["Counter_Value"]) for r in rows
                if r.get("Counter_Name") == name and "demod_kernel" in r.get("Kernel_Name", "")]
        if vals:
            out[name] = {"launches": len(vals), "avg_raw": sum(vals) / len(vals),
                         "min_raw": min(vals), "max_raw": max(vals)}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(base, "prof_summary.json"), "w"), indent=1)
