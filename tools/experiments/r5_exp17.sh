#!/bin/bash
# second hint level also for 4 - 5 KiB rounds (k15) against k14 (6 KiB and more)
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp17.txt
for spec in "--bauds 1200" "--bauds 12000" "--bauds 6000" "--bauds 2400" "--bauds 600" "--bauds 300" "--bauds 1500" "--bauds 150" "--bauds 300,1200,2400" "--bauds 1200 --streams 4096 --reps 40"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 $T/libafsk_k14.so $T/libafsk_k15.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp17.txt
done
