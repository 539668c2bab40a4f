#!/bin/bash
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp6.txt
for spec in "--bauds 2400" "--bauds 6000" "--bauds 12000" "--bauds 4000" "--bauds 3000" "--bauds 160" "--bauds 1200" "--bauds 300" "--bauds 800" \
            "--bauds 300,1200,2400" "--bauds 375,160,96,1200"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 8 $T/libafsk_r4.so $T/libafsk_k4.so $T/libafsk_k6.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp6.txt
done
( timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) | tee gpurun_out/r5_exp6_pytest.log
