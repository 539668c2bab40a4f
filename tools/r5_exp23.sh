#!/bin/bash
# timing-only variant without the decoded-byte stores (outputs differ by construction): do the mid-stream flush stores --
# which count in the same in-order vmcnt chain as the ring requests -- hold the short-symbol rates back on some boxes?
cd "$(dirname "$0")/.."
T=tools
rm -f gpurun_out/r5_exp23.txt
for spec in "--bauds 12000" "--bauds 6000" "--bauds 4000" "--bauds 1200" "--bauds 12000 --rounds 6"; do
  timeout -k 10 300 python tools/lib_ab.py --rounds 10 $spec $T/libafsk_k21.so $T/libafsk_k22x.so ${EXTRA_LIBS:-} 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp23.txt
done
