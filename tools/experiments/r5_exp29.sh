#!/bin/bash
# 200 baud (bit_frames 240) reads 0.595 in the final rates table: which r5 step did that?
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp29.txt
for spec in "--bauds 200" "--bauds 240" "--bauds 150"; do
  timeout -k 10 400 python tools/lib_ab.py --rounds 6 $spec $T/libafsk_k19.so $T/libafsk_k20.so $T/libafsk_k21.so $T/libafsk_k23.so $T/libafsk_k27.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp29.txt
done
