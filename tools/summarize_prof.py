#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + FETCH_SIZE / WRITE_SIZE PMC passes)
into a small JSON summary (gpurun_out/prof_summary_<name>.json) that is then committed
under profiles/.

    summarize_prof.py <gpurun_out> <workload> [timed_steps] [--kernel SUBSTR] [--name NAME]

<workload> names the rocprofv3 output directories (prof_trace_<workload>, prof_pmc1_<workload>,
prof_pmc2_<workload>); --kernel picks the kernel by a substring of its name (default "demod":
the demod kernels; "modulate_kernel", "block_amp_kernel" ... for the SURVEY 8(f) rows);
--name is the summary's own name (default: the workload).  With timed_steps,
"kernel_timed_region" covers the last timed_steps launches of the trace (bench.py's timed regions;
the launches before them are its clock pre-roll and warm-up).  The hipcc resource remark
(-Rpass-analysis=kernel-resource-usage) of the shipped kernel is recorded next to rocprofv3's
own register columns (which count allocation granules, not the compiler's figure)."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

args = [a for a in sys.argv[1:]]
kern = "demod"
name = None
if "--kernel" in args:
    i = args.index("--kernel"); kern = args[i + 1]; del args[i:i + 2]
if "--name" in args:
    i = args.index("--name"); name = args[i + 1]; del args[i:i + 2]
base = args[0] if len(args) > 0 else "gpurun_out"
wl = args[1] if len(args) > 1 else "config2"
timed_steps = int(args[2]) if len(args) > 2 else 0
name = name or wl
out = {"workload": wl, "kernel_filter": kern}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    return sorted(glob.glob(os.path.join(base, d + "_" + wl, "**", pat), recursive=True))


for f in find("prof_trace", "*kernel_stats.csv"):
    out["kernel_stats"] = list(csv.DictReader(open(f)))[:8]
    out["kernel_stats_file"] = f
kernel_names = set()
for f in find("prof_trace", "*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if kern in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    kernel_names |= {r["Kernel_Name"] for r in rows}
    if timed_steps and len(rows) >= timed_steps:
        t = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[-timed_steps:])
        out["kernel_timed_region"] = {"launches": len(t), "avg_ns": sum(t) / len(t),
                                      "median_ns": t[len(t) // 2], "min_ns": t[0], "max_ns": t[-1]}
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    if d:
        out["kernel_trace"] = {"launches": len(d), "avg_ns": sum(d) / len(d),
                               "median_ns": d[len(d) // 2], "min_ns": d[0], "max_ns": d[-1]}
        out["rocprof_resource_columns"] = {k: rows[0].get(k) for k in
                                           ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                            "Workgroup_Size_X", "Grid_Size_X", "Scratch_Size")}
out["kernel_names"] = sorted(kernel_names)
for cname, d in (("FETCH_SIZE", "prof_pmc1"), ("WRITE_SIZE", "prof_pmc2")):
    for f in find(d, "*counter_collection.csv"):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r.get("Counter_Name") == cname and kern in r.get("Kernel_Name", "")]
        if vals:
            out[cname] = {"launches": len(vals), "avg_raw_kib": sum(vals) / len(vals),
                          "min_raw_kib": min(vals), "max_raw_kib": max(vals)}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
    # exactly half the bytes of a wide (16 B/lane) coalesced streaming read -> double it.
    out["hbm_bytes_per_launch"] = int(2 * out["FETCH_SIZE"]["avg_raw_kib"] * 1024
                                      + out["WRITE_SIZE"]["avg_raw_kib"] * 1024)
    out["hbm_bytes_note"] = "2 x FETCH_SIZE (gfx950 halves wide coalesced reads) + WRITE_SIZE, KiB -> bytes, per launch"


def compiler_resources(kernel_names):
    """hipcc's own resource remark for the shipped kernels named in the trace (cross-compiles here:
    no GPU needed).  Only for the demod kernels, whose translation units are known."""
    res = {}
    csrc = os.path.join(ROOT, "afskmodem_amd", "csrc")
    jobs = set()
    for kn in kernel_names:
        m = re.search(r"demod_uniform_kernel_t<(\d+)", kn)
        if m:
            jobs.add(("afsk_demod_uniform.hip", f"-DAFSK_UNIFORM_BF={m.group(1)}"))
        elif "demod_kernel_t" in kn:
            jobs.add(("afsk_demod_big.hip" if re.search(r"true\s*>", kn) else "afsk_demod_small.hip", ""))
    for src, define in sorted(jobs):
        cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
               "-mllvm", "-structurizecfg-skip-uniform-regions",       # as afskmodem_amd/csrc/build.sh
               "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src] + ([define] if define else [])
        try:
            txt = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True, timeout=600).stderr
        except Exception as exc:  # noqa: BLE001
            res[src + " " + define] = f"hipcc not runnable here: {exc}"
            continue
        cur = None
        for ln in txt.splitlines():
            m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                          r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", ln)
            if not m:
                continue
            if m.group(1) == "Function Name":
                cur = res.setdefault(m.group(2), {})
            elif cur is not None:
                cur[m.group(1)] = int(m.group(2))
    return res


if kern == "demod" and kernel_names:
    out["compiler_resource_remark"] = compiler_resources(kernel_names)
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash: ties the figures to the kernel source they were measured on)
out["kernel_source_hash"] = bench.kernel_source_hash()
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(base, f"prof_summary_{name}.json"), "w"), indent=1)
# profiles/traffic_latest.json is what bench.py reports as roofline.traffic (with its provenance)
if kern == "demod" and "hbm_bytes_per_launch" in out and wl in bench.WORKLOADS and wl != "custom":
    tf = os.path.join(base, "traffic_latest.json")
    try:
        tj = json.load(open(tf))
    except Exception:  # noqa: BLE001
        tj = {}
    if tj.get("kernel_source_hash") != out["kernel_source_hash"]:
        tj = {"kernel_source_hash": out["kernel_source_hash"], "entries": {}}
    grid = int(out.get("rocprof_resource_columns", {}).get("Grid_Size_X") or 0)
    tj["entries"][wl] = {"streams": grid // 64 if grid else bench.WORKLOADS[wl][0],
                         "hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
                         "source": f"profiles/{os.environ.get('PROF_TAG', 'r3')}_{wl}_summary.json"}
    json.dump(tj, open(tf, "w"), indent=1)
