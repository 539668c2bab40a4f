#!/bin/bash
# control (the same r4 library under two names) + k3 / k4 through tools/lib_ab.py (one process, rotating order)
cd "$(dirname "$0")/../.."
T=tools
for spec in "--bauds 160" "--bauds 375" "--bauds 6000" "--bauds 12000" "--bauds 1200" "--bauds 750" "--bauds 96" \
            "--bauds 300,1200,2400" "--bauds 375,160,96,1200" "--bauds 375,160,96,1200 --streams 4096 --reps 40" \
            "--bauds 375,160,96,1200 --entry mixed"; do
  timeout -k 10 300 python tools/lib_ab.py $spec $T/libafsk_r4.so $T/libafsk_r4copy.so $T/libafsk_k2.so $T/libafsk_k3.so $T/libafsk_k4.so 2>&1 | grep -v "^bench.py\|Warning\|warn" | tee -a gpurun_out/r5_exp4.txt
done
