#!/bin/bash
# r6 visit 9: afsk_gate_batch_slots -- whole GPU suite, the chain row, then the profile pass (the kernel source hash moved
# with afsk_kernels.h: traffic_latest.json must be re-measured)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) | tee gpurun_out/r6_v9_suite.txt
grep -q failed gpurun_out/r6_v9_suite.txt && exit 1
timeout -k 10 300 python bench.py --sub f2_chain,f2_gate --steps 5 --warmup 2 --no-cpu-baseline --next-reps 20 2>/dev/null | tail -1 > gpurun_out/r6_v9_chain.json
python -c "
import json; d = json.load(open('gpurun_out/r6_v9_chain.json')); print(json.dumps(d['sub_records']))"
SKIP_TESTS=1 PROF_ONLY=1 bash tools/gpu_round.sh prof r6 > gpurun_out/r6_prof_visit.log 2>&1; tail -2 gpurun_out/r6_prof_visit.log
