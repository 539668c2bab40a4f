#!/bin/bash
# does the ADDRESS ORDER of a walk matter?  One rate (1200 baud, uniform kernel, results in stream order either way);
# stream s reads the samples of stream (s % K) * (n / K) + s / K: the 2048 streams in flight span K times the range
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp37.txt
for k in 0 4 18 64; do
  timeout -k 10 400 python tools/lib_ab.py --rounds 8 --bauds 1200 --offset-transpose $k $T/libafsk_k30.so $T/libafsk_k21.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids\|streams differ" | sed "s/^bauds/transpose $k: bauds/" | tee -a gpurun_out/r5_exp37.txt
done
