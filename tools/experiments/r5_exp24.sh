#!/bin/bash
# what the decoded-byte stores cost and which form is cheapest.  k21: byte stores in the middle of the stream; k22: bytes
# wait in LDS, byte stores at the end of the stream, then vmcnt(0); k23a: the same but the ring drain (vmcnt(0)) BEFORE the
# stores (the wave ends without waiting for their acknowledgement); k23b: dword stores, acknowledged; k23: drain first +
# dword stores; k22x / k23x: timing bounds without the byte stores / without any output store (outputs differ).
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp24.txt
for spec in "--bauds 12000" "--bauds 6000" "--bauds 3000" "--bauds 1200" "--bauds 300" "--bauds 1200 --streams 4096 --reps 40" "--bauds 375,160,96,1200"; do
  timeout -k 10 400 python tools/lib_ab.py --rounds 8 $spec $T/libafsk_k21.so $T/libafsk_k22.so $T/libafsk_k23a.so $T/libafsk_k23b.so $T/libafsk_k23.so $T/libafsk_k22x.so $T/libafsk_k23x.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp24.txt
done
