#!/bin/bash
# k5 (k4 + phase-C cuts + watermark fast paths) against r4 / k4, then the whole GPU suite on the in-tree library (= k5)
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp5.txt
for spec in "--bauds 160" "--bauds 375" "--bauds 6000" "--bauds 12000" "--bauds 1200" "--bauds 800" "--bauds 96" "--bauds 300" "--bauds 2400" "--bauds 3000" "--bauds 100" \
            "--bauds 300,1200,2400" "--bauds 375,160,96,1200" "--bauds 375,160,96,1200 --streams 4096 --reps 40" "--bauds 1200 --streams 4096 --reps 40"; do
  timeout -k 10 300 python tools/lib_ab.py $spec $T/libafsk_r4.so $T/libafsk_k4.so $T/libafsk_k5.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp5.txt
done
( timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) | tee gpurun_out/r5_exp5_pytest.log
