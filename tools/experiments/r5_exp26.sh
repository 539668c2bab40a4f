#!/bin/bash
# XCD-aware block -> streams mapping, local form (k25: inside every group of 64 consecutive blocks the 8 blocks of one XCD
# take 8 consecutive blocks of streams = one line of every status array; the launch still walks the input front to back)
# against blockIdx order (k23) and the state before the output buffer (k21); k23x = no output stores at all (timing bound)
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp26.txt
for spec in "--bauds 12000" "--bauds 6000" "--bauds 3000" "--bauds 1200" "--bauds 300" "--bauds 160" "--bauds 1200 --streams 4096 --reps 40" "--bauds 375,160,96,1200" "--bauds 300,1200,2400" "--bauds 375,160,96,1200 --streams 4096 --reps 40" "--bauds 1200 --entry mixed"; do
  timeout -k 10 400 python tools/lib_ab.py --rounds 8 $spec $T/libafsk_k21.so $T/libafsk_k23.so $T/libafsk_k24.so $T/libafsk_k25.so $T/libafsk_k23x.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp26.txt
done
