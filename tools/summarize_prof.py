#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + FETCH_SIZE / WRITE_SIZE PMC passes)
into a small JSON summary (gpurun_out/prof_summary_<workload>.json) that is then committed
under profiles/.  Usage: summarize_prof.py <gpurun_out> <workload> [timed_steps]
With timed_steps, "demod_kernel_timed_region" covers the last timed_steps launches of the trace
(bench.py's timed region; the launches before it are its clock pre-roll and warm-up)."""
import csv
import glob
import json
import os
import sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
wl = sys.argv[2] if len(sys.argv) > 2 else "config2"
timed_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
out = {"workload": wl}


def find(d, pat):
    return sorted(glob.glob(os.path.join(base, d + "_" + wl, "**", pat), recursive=True))


for f in find("prof_trace", "*kernel_stats.csv"):
    out["kernel_stats"] = list(csv.DictReader(open(f)))[:6]
    out["kernel_stats_file"] = f
for f in find("prof_trace", "*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "demod_kernel" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if timed_steps and len(rows) >= timed_steps:
        t = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[-timed_steps:])
        out["demod_kernel_timed_region"] = {"launches": len(t), "avg_ns": sum(t) / len(t),
                                            "median_ns": t[len(t) // 2], "min_ns": t[0],
                                            "max_ns": t[-1]}
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    if d:
        out["demod_kernel_trace"] = {"launches": len(d), "avg_ns": sum(d) / len(d),
                                     "median_ns": d[len(d) // 2], "min_ns": d[0], "max_ns": d[-1]}
        out["demod_kernel_resources"] = {k: rows[0].get(k) for k in
                                         ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                          "Workgroup_Size_X", "Grid_Size_X", "Scratch_Size")}
for name, d in (("FETCH_SIZE", "prof_pmc1"), ("WRITE_SIZE", "prof_pmc2")):
    for f in find(d, "*counter_collection.csv"):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r.get("Counter_Name") == name and "demod_kernel" in r.get("Kernel_Name", "")]
        if vals:
            out[name] = {"launches": len(vals), "avg_raw_kib": sum(vals) / len(vals),
                         "min_raw_kib": min(vals), "max_raw_kib": max(vals)}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
    # exactly half the bytes of a wide (16 B/lane) coalesced streaming read -> double it.
    out["hbm_bytes_per_launch"] = int(2 * out["FETCH_SIZE"]["avg_raw_kib"] * 1024
                                      + out["WRITE_SIZE"]["avg_raw_kib"] * 1024)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_hash: ties the figures to the kernel source they were measured on)
out["kernel_source_hash"] = bench.kernel_source_hash()
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(base, f"prof_summary_{wl}.json"), "w"), indent=1)
# profiles/traffic_latest.json is what bench.py reports as roofline.traffic (with its provenance)
if "hbm_bytes_per_launch" in out and wl in bench.WORKLOADS and wl != "custom":
    tf = os.path.join(base, "traffic_latest.json")
    try:
        tj = json.load(open(tf))
    except Exception:  # noqa: BLE001
        tj = {}
    if tj.get("kernel_source_hash") != out["kernel_source_hash"]:
        tj = {"kernel_source_hash": out["kernel_source_hash"], "entries": {}}
    grid = int(out.get("demod_kernel_resources", {}).get("Grid_Size_X") or 0)
    tj["entries"][wl] = {"streams": grid // 64 if grid else bench.WORKLOADS[wl][0],
                         "hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
                         "source": f"profiles/{os.environ.get('PROF_TAG', 'r2')}_{wl}_summary.json"}
    json.dump(tj, open(tf, "w"), indent=1)
