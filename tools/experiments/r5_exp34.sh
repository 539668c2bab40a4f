#!/bin/bash
# rows written through (sc0 sc1: k32w) or nt (k32n) from the LDS output buffer at the end of the stream, status words
# plain (they merge in the L2) -- against the shipped library (k30: plain byte stores from the flushes); rows on lines
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp34.txt
for spec in "--bauds 12000" "--bauds 6000" "--bauds 4000" "--bauds 3000" "--bauds 1200" "--bauds 375,160,96,1200"; do
  timeout -k 10 400 python tools/lib_ab.py --rounds 8 $spec $T/libafsk_k30.so $T/libafsk_k32w.so $T/libafsk_k32n.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp34.txt
done
