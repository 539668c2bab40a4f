#!/bin/bash
# f2 gate: block_amp_kernel with the XCD-local permutation of its workgroups (k30) against blockIdx order (k29): the 32
# amplitudes of a 128-byte line come from 8 workgroups
cd "$(dirname "$0")/../.."
rm -f gpurun_out/r5_exp32.txt
for lib in k29 k30 k29 k30 k29 k30; do
  AFSK_AMD_LIB=$PWD/tools/libafsk_$lib.so timeout -k 10 300 python bench.py --sub f2_gate --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open(l['full_record']))
g=f['sub_records']['f2_gate']['captures_65536']
print('$lib', 'gate frac', g['roofline']['frac'], 'kernel_ms', g['roofline']['kernel_ms'], 'bursts', g['bursts_found'], '| headline', l['roofline']['frac'])
" | tee -a gpurun_out/r5_exp32.txt
done
( timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gate or listen" 2>&1 | tail -2 ) | tee -a gpurun_out/r5_exp32.txt
