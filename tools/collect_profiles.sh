#!/bin/bash
# Copy what one `tools/gpu_round.sh prof <tag>` visit left under gpurun_out/ (scratch) into profiles/
# (tracked): per workload the rocprofv3 summary, the kernel stats CSV, the launch timeline and the bench
# line printed under the profiler; the plain bench line of the visit; traffic_latest.json.
#   bash tools/collect_profiles.sh <tag> [workload ...]
cd "$(dirname "$0")/.."
T=${1:-r3}; shift || true
WL=${*:-config5 config2 config3 custom100 custom150 custom200}
for w in $WL; do
  [ -f gpurun_out/prof_summary_$w.json ] && cp gpurun_out/prof_summary_$w.json profiles/${T}_${w}_summary.json
  f=$(ls gpurun_out/prof_trace_$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/${T}_${w}_kernel_stats.csv
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
