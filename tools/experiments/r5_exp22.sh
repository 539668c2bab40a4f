#!/bin/bash
# ring refill: four LDS-DMA requests per M0 / scalar offset through the instruction's immediate offset (issue_run)
# against the same build with one M0 + offset per request; parity first (the immediate must move the LDS address too)
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp22.txt
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 ) | tee gpurun_out/r5_exp22_pytest.log
grep -q " passed" gpurun_out/r5_exp22_pytest.log && ! grep -q "failed" gpurun_out/r5_exp22_pytest.log || exit 1
for spec in "--bauds 12000" "--bauds 6000" "--bauds 4000" "--bauds 3000" "--bauds 2000" "--bauds 1000" "--bauds 2400" "--bauds 1200" "--bauds 600" "--bauds 300" "--bauds 500" "--bauds 160" "--bauds 40" "--bauds 1200 --streams 4096 --reps 40" "--bauds 300,1200,2400" "--bauds 375,160,96,1200" "--bauds 375,160,96,1200 --streams 4096 --reps 40"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 $T/libafsk_k20.so $T/libafsk_k21.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp22.txt
done
