#!/bin/bash
# per-stream kernel: tail hint also for bit_frames 4 / 8 (k17) against k16
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp18.txt
M18="12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240"
for spec in "--bauds $M18" "--bauds $M18 --streams 4096 --reps 40" "--bauds 300,1200,2400" "--bauds 375,160,96,1200" "--bauds 12000 --entry mixed" "--bauds 6000 --entry mixed" "--bauds 1200 --entry mixed" "--bauds 12000,6000,1200,300"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 8 $T/libafsk_k16.so $T/libafsk_k17.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp18.txt
done
