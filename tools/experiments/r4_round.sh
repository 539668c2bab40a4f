#!/bin/bash
# One GPU visit of round 4: parity suite, smoke, A/B of the grouped dispatch, the bench line.  Outputs: gpurun_out/r4_*.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
T=${1:-r4}
set -o pipefail
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${T}_pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
B="--sub '' --no-cpu-baseline"
for args in "--workload config3" "--workload config3 --entry mixed" "--workload custom --bauds 375,160,96,1200" "--workload custom --bauds 375,160,96,1200 --entry mixed" "--workload custom --bauds 375,160,96,1200 --streams 65536" "--workload config2 --steps 200"; do
  echo "== $args"
  timeout -k 10 300 python bench.py $args --sub "" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'entry', 'roundtrip_match_rate')}, d['roofline']['frac'], d['roofline']['kernel_ms'])"
done
