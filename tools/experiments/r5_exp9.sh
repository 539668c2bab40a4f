#!/bin/bash
# multi_rounds: decisions of the first slices before the refill (EARLY = 0 / 1 / SPL/2) -- one process per rate
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp9.txt
for spec in "--bauds 12000" "--bauds 6000" "--bauds 4000" "--bauds 3000" "--bauds 2000" "--bauds 6000 --streams 4096 --reps 40" "--bauds 4000 --streams 16384 --reps 10"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 $T/libafsk_k8.so $T/libafsk_k9zero.so $T/libafsk_k9one.so $T/libafsk_k9half.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp9.txt
done
