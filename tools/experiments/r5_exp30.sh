#!/bin/bash
# is the LDS output buffer (k22 / k23: bytes wait for the end of the stream, dword stores behind the ring drain) worth its
# code once the XCD permutation is in?  k28 = k21 + the permutation only (byte stores in the middle of the stream);
# k27 = shipped (buffer + permutation); k21 = neither
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp30.txt
for spec in "--bauds 1200" "--bauds 12000" "--bauds 6000" "--bauds 3000" "--bauds 300" "--bauds 160" "--bauds 1200 --streams 4096 --reps 40" "--bauds 375,160,96,1200" "--bauds 300,1200,2400" "--bauds 375,160,96,1200 --streams 4096 --reps 40" "--bauds 12000 --entry mixed"; do
  timeout -k 10 400 python tools/lib_ab.py --rounds 8 $spec $T/libafsk_k21.so $T/libafsk_k28.so $T/libafsk_k27.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp30.txt
done
