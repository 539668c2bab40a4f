#!/bin/bash
# r6 visit 4: re-base by whole dwords + ODD round forms: lead tests (incl. ragged), per-shift table, exp1, chain + ragged rows
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout -k 10 900 python -m pytest tests/test_gpu_lead.py -x -q 2>&1 | tail -15 ) | tee gpurun_out/r6_v4_lead_tests.log
grep -q "passed" gpurun_out/r6_v4_lead_tests.log && ! grep -q "failed" gpurun_out/r6_v4_lead_tests.log || exit 1
LEADS="0 1 2 3 8" bash tools/experiments/r6_exp3_leads.sh
cp gpurun_out/r6_exp3.txt gpurun_out/r6_exp3_v4.txt
bash tools/experiments/r6_exp1_lead_baseline.sh
cp gpurun_out/r6_exp1.txt gpurun_out/r6_exp1_v4.txt
timeout -k 10 600 python bench.py --sub f2_chain,ragged_lengths --steps 5 --warmup 2 --no-cpu-baseline --next-reps 10 2>gpurun_out/r6_v4_rows.err | tail -1 > gpurun_out/r6_v4_rows.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full_n1_partial.json")) if False else None
PY
ls gpurun_out/bench_full* | tail -3
