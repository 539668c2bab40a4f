#!/bin/bash
# Copy what one `tools/gpu_round.sh prof <tag>` visit left under gpurun_out/ (scratch) into profiles/
# (tracked): per workload the rocprofv3 summary, the kernel stats CSV, the launch timeline and the bench
# line printed under the profiler; the plain bench line of the visit; traffic_latest.json.
#   bash tools/collect_profiles.sh <tag> [workload ...]
cd "$(dirname "$0")/.."
T=${1:-r6}; shift || true
WL=${*:-config5 config5_lead config2 config3}
for w in $WL; do
  # ONE rocprofv3 run per summary: the committed CSV is the very file the summary was computed from (its
  # `kernel_stats_file`; gpurun_out/ accumulates the directories of earlier visits, so never "the first CSV
  # found"), and the committed summary names the committed copy and its sha256
  [ -f gpurun_out/prof_summary_$w.json ] && python - "$T" "$w" <<'PY'
import hashlib, json, os, shutil, sys
T, w = sys.argv[1:3]
d = json.load(open(f"gpurun_out/prof_summary_{w}.json"))
src = d.get("kernel_stats_file")
if src and os.path.exists(src):
    dst = f"profiles/{T}_{w}_kernel_stats.csv"
    shutil.copyfile(src, dst)
    d["kernel_stats_file_on_the_gpu_box"] = src
    d["kernel_stats_file"] = dst
    d["kernel_stats_sha256"] = hashlib.sha256(open(dst, "rb").read()).hexdigest()
json.dump(d, open(f"profiles/{T}_{w}_summary.json", "w"), indent=1)
PY
  [ -s gpurun_out/timeline_$w.txt ] && cp gpurun_out/timeline_$w.txt profiles/${T}_${w}_timeline.txt
  [ -s gpurun_out/prof_bench_$w.json ] && cp gpurun_out/prof_bench_$w.json profiles/${T}_bench_under_rocprof_${w}.json
done
for k in modulate gate gate_scan; do
  [ -f gpurun_out/prof_summary_$k.json ] && cp gpurun_out/prof_summary_$k.json profiles/${T}_${k}_summary.json
done
f=$(ls gpurun_out/prof_trace_next/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" profiles/${T}_next_rows_kernel_stats.csv
[ -s gpurun_out/prof_bench_next.json ] && cp gpurun_out/prof_bench_next.json profiles/${T}_bench_under_rocprof_next_rows.json
[ -s gpurun_out/${T}_bench_main.json ] && cp gpurun_out/${T}_bench_main.json profiles/${T}_bench_n1.json
[ -f gpurun_out/traffic_latest.json ] && cp gpurun_out/traffic_latest.json profiles/traffic_latest.json
ls profiles | grep "^${T}_" | wc -l
