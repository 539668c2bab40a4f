#!/bin/bash
# r6 exp6: the r5 library (tools/libafsk_r5.so, built from commit 7c6cb12) against the r6 library in ONE process,
# interleaved launches on one resident batch (tools/lib_ab.py): clean streams (did the re-base cost the aligned path
# anything?) and led-in streams (what it bought), outputs compared
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r6_exp6.txt
: > $O
A=afskmodem_amd/csrc/libafsk_amd.so; B=tools/libafsk_r5.so
for spec in "--bauds 1200" "--bauds 800" "--bauds 480" "--bauds 2400" "--bauds 12000" "--bauds 160" "--bauds 300,1200,2400" "--bauds 375,160,96,1200" "--bauds 1200 --streams 4096 --reps 40" "--bauds 1200 --lead random" "--bauds 2400 --lead random" "--bauds 300 --lead random" "--bauds 12000 --lead random" "--bauds 300,1200,2400 --lead random" "--bauds 1200 --lead random --streams 4096 --reps 40"; do
  echo "== $spec" >> $O
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 $A $B 2>&1 | grep -v "amdgpu.ids\|^bench.py\|Warning\|warn" | tail -3 >> $O
done
cat $O
