#!/bin/bash
# r6 visit 3: the whole GPU suite on the re-based ring, then exp1 again (after)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout -k 10 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) | tee gpurun_out/r6_v3_pytest_gpu.log
grep -q "failed" gpurun_out/r6_v3_pytest_gpu.log && exit 1
bash tools/experiments/r6_exp1_lead_baseline.sh
cp gpurun_out/r6_exp1.txt gpurun_out/r6_exp1_after_rebase.txt
