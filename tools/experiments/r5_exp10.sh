#!/bin/bash
# chunks per round for bit_frames 8 / 12 / 16 / 24: library (5 / 6 / 6 / 6) against rA (6 / 9 / 8 / 9) and rB (8 / 9 / 8 / 9)
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp10.txt
for spec in "--bauds 3000" "--bauds 4000" "--bauds 2000" "--bauds 6000" "--bauds 3000 --streams 4096 --reps 40" "--bauds 4000 --streams 4096 --reps 40" "--bauds 6000 --streams 4096 --reps 40"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 afskmodem_amd/csrc/libafsk_amd.so $T/libafsk_rA.so $T/libafsk_rB.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp10.txt
done
