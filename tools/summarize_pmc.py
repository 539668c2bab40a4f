#!/usr/bin/env python3
"""Average of every counter of rocprofv3 --pmc passes over the dispatches of one kernel.
   summarize_pmc.py "<rocprofv3 output dir> [<dir> ...]" [kernel substring = demod]  ->  JSON on stdout
Several directories = several passes of the same command with different counter sets (8 SQ slots per pass)."""
import csv
import glob
import json
import os
import sys

dirs = sys.argv[1].split()
kern = sys.argv[2] if len(sys.argv) > 2 else "demod"
acc, names = {}, set()
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern not in r.get("Kernel_Name", ""):
                continue
            names.add(r["Kernel_Name"])
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {"kernel_filter": kern, "kernel_names": sorted(names), "passes": dirs,
       "counters": {k: {"dispatches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(acc.items())}}
c = out["counters"]
avg = lambda k: c[k]["avg"]  # noqa: E731
if "SQ_WAVE_CYCLES" in c:
    wc = avg("SQ_WAVE_CYCLES")
    out["share_of_wave_cycles"] = {k: round(avg(k) / wc, 4) for k in
                                   ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                    "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM") if k in c}
if "SQ_WAVES" in c and avg("SQ_WAVES") > 0:
    w = avg("SQ_WAVES")
    out["per_wave"] = {k.replace("SQ_INSTS_", "").replace("SQ_", "").lower(): round(avg(k) / w, 1) for k in
                       ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM",
                        "SQ_IFETCH") if k in c}
    if "SQ_WAVE_CYCLES" in c:
        out["per_wave"]["wave_cycles_x4"] = round(4 * avg("SQ_WAVE_CYCLES") / w, 0)   # quad-cycles -> cycles
if "SQ_LDS_IDX_ACTIVE" in c and avg("SQ_LDS_IDX_ACTIVE") > 0:
    la = avg("SQ_LDS_IDX_ACTIVE")
    out["share_of_lds_active_cycles"] = {k: round(avg(k) / la, 4) for k in
                                         ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_UNALIGNED_STALL") if k in c}
if "SQC_ICACHE_REQ" in c and avg("SQC_ICACHE_REQ") > 0:
    out["icache"] = {"req": avg("SQC_ICACHE_REQ"), "miss_rate": round(avg("SQC_ICACHE_MISSES") / avg("SQC_ICACHE_REQ"), 5)
                     if "SQC_ICACHE_MISSES" in c else None}
print(json.dumps(out, indent=1))
