#!/bin/bash
# grouped walk in bucket order inside WINDOWS of consecutive streams (AFSK_GROUP_WINDOW; 0 = over the whole batch, r4's
# form): the streams in flight then stay within a window's address range -- rates cycling over the streams; all window
# sizes in ONE process on one resident batch (separate processes differ by 2 - 4 % through their allocations alone)
cd "$(dirname "$0")/../.."
L=tools/libafsk_k34.so
rm -f gpurun_out/r5_exp38.txt
for spec in "--bauds 375,160,96,1200" "--bauds 12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240" "--bauds 375,160,96,1200 --streams 4096 --reps 40" "--bauds 300,1200,2400,600"; do
  timeout -k 10 500 python tools/lib_ab.py --rounds 8 $spec $L@0 $L@512 $L@1024 $L@2048 $L@4096 $L@8192 $L@16384 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | cut -c1-170 | tee -a gpurun_out/r5_exp38.txt
done
