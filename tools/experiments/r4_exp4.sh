#!/bin/bash
# where does the tail hint start to pay for bit_frames 4 / 8 (uniform kernels)?  current library (hint from 8192 streams) vs none
cd "$(dirname "$0")/../.."
R=$(pwd)
for rep in 1 2; do
for n in 8192 12288 16384 32768; do
  for b in 12000 6000; do
    for lib in afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_nohintshort.so; do
      AFSK_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py --workload custom --bauds $b --streams $n --steps 100 --sub "" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b x $n $(basename $lib)', d['roofline']['frac'], d['roofline']['kernel_ms'])"
    done
  done
done
done
