#!/bin/bash
# decoded bytes wait in LDS for the end of the stream (k22) against stores in the middle of the stream (k21): a store sits
# in the same in-order vmcnt chain as the ring requests.  k22x = k21 without any byte store (timing bound; outputs differ).
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp23.txt
for spec in "--bauds 12000" "--bauds 6000" "--bauds 4000" "--bauds 3000" "--bauds 2400" "--bauds 1200" "--bauds 300" "--bauds 1200 --streams 4096 --reps 40" "--bauds 375,160,96,1200" "--bauds 12000 --entry mixed"; do
  timeout -k 10 300 python tools/lib_ab.py --rounds 10 $spec $T/libafsk_k21.so $T/libafsk_k22.so $T/libafsk_k22x.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp23.txt
done
