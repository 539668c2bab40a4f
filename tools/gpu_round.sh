#!/bin/bash
# One GPU visit: parity tests, smoke, bench, rocprof kernel trace + PMC. Outputs under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -60 ) | tee gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) | tee gpurun_out/smoke.log
( timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 ) | tee gpurun_out/bench.log
if [ "${1:-}" = "prof" ]; then
  rm -rf gpurun_out/prof_trace gpurun_out/prof_pmc1 gpurun_out/prof_pmc2
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_trace -- python3 /root/repo/bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -3 )
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/prof_pmc1 -- python3 /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -2 )
  ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/prof_pmc2 -- python3 /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -2 )
  find gpurun_out/prof_trace gpurun_out/prof_pmc1 gpurun_out/prof_pmc2 -type f | head -30
  python tools/summarize_prof.py gpurun_out 2>&1 | tail -30
fi
