#!/bin/bash
# output rows on 128-byte lines: bench.py's stride (out_stride_for: a multiple of 4) against the same rounded up to a
# multiple of 128 -- every row then starts on a line of its own and covers ceil(nbytes / 128) lines instead of one more
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp31.txt
for baud in 12000 6000 4000 3000 2400 2000 1500 1200; do
  for al in 0 128; do
    timeout -k 10 400 python tools/lib_ab.py --rounds 8 --bauds $baud --stride-align $al $T/libafsk_k29.so $T/libafsk_k21.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | tee -a gpurun_out/r5_exp31.txt
  done
done
