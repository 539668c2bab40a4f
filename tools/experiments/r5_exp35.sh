#!/bin/bash
# grouped walk, results at the original stream numbers: rates cycling per stream (a status line holds every rate: it is
# written at as many different times) against rates in contiguous blocks of streams (the sorted walk = stream order:
# lines merge) -- the cost of the partial status lines of a rate-sorted walk
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp35.txt
for order in cycle blocks; do
  for spec in "--bauds 375,160,96,1200" "--bauds 12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240"; do
    echo "rate order: $order" | tee -a gpurun_out/r5_exp35.txt
    timeout -k 10 400 python tools/lib_ab.py --rounds 8 --rate-order $order $spec $T/libafsk_k30.so $T/libafsk_k21.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp35.txt
  done
done
