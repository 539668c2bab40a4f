#!/bin/bash
# k13: probes closer than a round only for rounds of 6 KiB and more (+ partial rounds), against k11
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp15.txt
for spec in "--bauds 1200" "--bauds 12000" "--bauds 6000" "--bauds 2400" "--bauds 300" "--bauds 4000" "--bauds 3000" "--bauds 2000" "--bauds 1000" "--bauds 750" "--bauds 800" "--bauds 500" "--bauds 375" "--bauds 160" "--bauds 96" "--bauds 200" "--bauds 300,1200,2400" "--bauds 375,160,96,1200"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 8 $T/libafsk_k11.so $T/libafsk_k13.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp15.txt
done
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 ) | tee gpurun_out/r5_exp15_pytest.log
