#!/usr/bin/env python3
"""Average of every counter of a rocprofv3 --pmc pass over the dispatches of one kernel.
   summarize_pmc.py <rocprofv3 output dir> [kernel substring = demod]  ->  JSON on stdout"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "demod"
acc, names = {}, set()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern not in r.get("Kernel_Name", ""):
            continue
        names.add(r["Kernel_Name"])
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {"kernel_filter": kern, "kernel_names": sorted(names),
       "counters": {k: {"dispatches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(acc.items())}}
c = out["counters"]
if "SQ_WAVE_CYCLES" in c:
    wc = c["SQ_WAVE_CYCLES"]["avg"]
    out["share_of_wave_cycles"] = {k: round(c[k]["avg"] / wc, 4) for k in
                                   ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in c}
if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]["avg"] > 0:
    la = c["SQ_LDS_IDX_ACTIVE"]["avg"]
    out["share_of_lds_active_cycles"] = {k: round(c[k]["avg"] / la, 4) for k in
                                         ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_UNALIGNED_STALL") if k in c}
print(json.dumps(out, indent=1))
