#!/bin/bash
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp13.txt
for spec in "--bauds 12000" "--bauds 6000" "--bauds 4000" "--bauds 1200" "--bauds 160" "--bauds 300" "--bauds 800" "--bauds 300,1200,2400" "--bauds 1200 --streams 4096 --reps 40"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 $T/libafsk_r4.so $T/libafsk_k10.so $T/libafsk_k11.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp13.txt
done
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 ) | tee gpurun_out/r5_exp13_pytest.log
