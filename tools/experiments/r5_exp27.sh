#!/bin/bash
# k26 = k25 + the unwritten remainder of an output row zero-filled where that completes 128-byte lines (no partial-line
# write-backs); k25 = local XCD permutation; k23 = blockIdx order; k23x = no output stores at all (timing bound)
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp27.txt
for spec in "--bauds 1200" "--bauds 12000" "--bauds 6000" "--bauds 3000" "--bauds 2400" "--bauds 300" "--bauds 160" "--bauds 1200 --streams 4096 --reps 40" "--bauds 375,160,96,1200" "--bauds 300,1200,2400" "--bauds 1200 --entry mixed"; do
  timeout -k 10 400 python tools/lib_ab.py --rounds 8 $spec $T/libafsk_k23.so $T/libafsk_k25.so $T/libafsk_k26.so $T/libafsk_k23x.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp27.txt
done
