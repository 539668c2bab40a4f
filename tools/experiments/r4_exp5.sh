#!/bin/bash
# chunks per round for bit_frames 4 / 8: R = 5 (library) against 4 and 6, uniform kernels, steady state and 4096 streams
cd "$(dirname "$0")/../.."
R=$(pwd)
for rep in 1 2; do
for n in 65536 4096; do
  for b in 12000 6000; do
    for lib in afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_r4short.so tools/libafsk_r6short.so; do
      AFSK_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py --workload custom --bauds $b --streams $n --steps $((n > 10000 ? 30 : 200)) --sub "" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b x $n $(basename $lib)', d['roofline']['frac'], d['roofline']['kernel_ms'], d['roundtrip_match_rate'])"
    done
  done
done
done
for lib in tools/libafsk_r4short.so tools/libafsk_r6short.so; do
  AFSK_AMD_LIB=$R/$lib timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_cases_device or large_launch or every_alignment or fuzz_noise" 2>&1 | tail -2
done
