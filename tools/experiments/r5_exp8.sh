#!/bin/bash
# k8 (SDWA compares at 12000 baud, tail hint for grouped launches from 4096 streams) against r4 / k7
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp8.txt
for spec in "--bauds 12000" "--bauds 12000 --streams 8192 --reps 20" "--bauds 375,160,96,1200 --streams 4096 --reps 40" "--bauds 375,160,96,1200 --streams 5000 --reps 40" \
            "--bauds 12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240 --streams 4096 --reps 40" \
            "--bauds 12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240" "--bauds 375,160,96,1200"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 8 $T/libafsk_r4.so $T/libafsk_k7.so $T/libafsk_k8.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp8.txt
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r5_exp8_pytest.log
