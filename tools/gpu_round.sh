#!/bin/bash
# One GPU visit: parity tests, smoke, bench lines for configs 2/3/4/5, rocprofv3 kernel trace
# + PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs).  Outputs under gpurun_out/.
#   bash tools/gpu_round.sh [prof]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) | tee gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) | tee gpurun_out/smoke.log
( timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -1 ) | tee gpurun_out/bench_config2.json
for w in config3 config4 config5; do
  ( timeout 900 python bench.py --workload $w --steps 20 --warmup 3 --cpu-sample-streams 1536 2>&1 | tail -1 ) | tee gpurun_out/bench_$w.json
done
if [ "${1:-}" = "prof" ]; then
  for w in config2 config3; do
    steps=200; [ $w = config3 ] && steps=20
    rm -rf gpurun_out/prof_trace_$w gpurun_out/prof_pmc1_$w gpurun_out/prof_pmc2_$w
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_trace_$w -- python3 /root/repo/bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 )
    ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/prof_pmc1_$w -- python3 /root/repo/bench.py --workload $w --steps 6 --warmup 2 --preroll-ms 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100 )
    ( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/prof_pmc2_$w -- python3 /root/repo/bench.py --workload $w --steps 6 --warmup 2 --preroll-ms 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100 )
    python tools/summarize_prof.py gpurun_out $w $steps 2>&1 | tail -40
    python tools/trace_timeline.py gpurun_out/prof_trace_$w 2>&1 | tee gpurun_out/timeline_$w.txt
  done
fi
