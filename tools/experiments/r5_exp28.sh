#!/bin/bash
# cache policy of the output stores (decoded bytes + status words): plain (k27 = k25 rebuilt), nt (k27n), sc0 sc1 =
# write-through (k27w); k23x = no output stores at all (timing bound)
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp28.txt
for spec in "--bauds 1200" "--bauds 12000" "--bauds 6000" "--bauds 300" "--bauds 160" "--bauds 1200 --streams 4096 --reps 40" "--bauds 375,160,96,1200"; do
  timeout -k 10 400 python tools/lib_ab.py --rounds 8 $spec $T/libafsk_k25.so $T/libafsk_k27.so $T/libafsk_k27n.so $T/libafsk_k27w.so $T/libafsk_k23x.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp28.txt
done
