#!/bin/bash
# One GPU visit: parity tests, smoke, bench, rocprof kernel trace. Outputs under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 ) | tee gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) | tee gpurun_out/smoke.log
( timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 ) | tee gpurun_out/bench.log
