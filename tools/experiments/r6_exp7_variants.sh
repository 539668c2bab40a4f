#!/bin/bash
# r6 exp7: micro-variants of the 1200-baud round against the shipped library, one process, interleaved (tools/lib_ab.py)
#   v1 = squelch amplitude of the first HALF symbol first (sum|x| is monotone: loud by half => loud), the rest only when
#        some lane is not loud by its half
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r6_exp7.txt
: > $O
A=afskmodem_amd/csrc/libafsk_amd.so
for spec in "--bauds 1200" "--bauds 1200 --lead random" "--bauds 1200 --streams 4096 --reps 40" "--bauds 1200" ; do
  echo "== $spec" >> $O
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 14 $A ${VARIANTS:-tools/libafsk_v1.so} $A 2>&1 | grep -v "amdgpu.ids\|^bench.py\|Warning\|warn" | tail -4 >> $O
done
cat $O
