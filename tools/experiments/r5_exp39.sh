#!/bin/bash
# with the windowed walk in, does the bucket order pay from fewer than four rates on?  config #3's rates (and two
# rates), plan sorted from 2 / 3 rates against stream order (sorted from 4: r4's rule); window 4096 and 2048
cd "$(dirname "$0")/../.."
L=tools/libafsk_k35.so
rm -f gpurun_out/r5_exp39.txt
for spec in "--bauds 300,1200,2400" "--bauds 1200,2400" "--bauds 300,1200,2400 --streams 4096 --reps 40"; do
  timeout -k 10 500 python tools/lib_ab.py --rounds 8 $spec $L@4096:4 $L@4096:2 $L@2048:2 $L@1024:2 $L@0:2 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | cut -c1-170 | tee -a gpurun_out/r5_exp39.txt
done
