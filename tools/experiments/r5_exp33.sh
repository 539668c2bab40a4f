#!/bin/bash
# f1 modulator with the XCD-local permutation of its workgroups (k31: an XCD writes 8 consecutive 16 KiB chunks) against
# blockIdx order (k30) -- full-line writes, so nothing to merge: is there a locality effect at all?
cd "$(dirname "$0")/../.."
rm -f gpurun_out/r5_exp33.txt
for lib in k30 k31 k30 k31 k30 k31; do
  AFSK_AMD_LIB=$PWD/tools/libafsk_$lib.so timeout -k 10 300 python bench.py --sub f1_modulate --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open(l['full_record']))
g=f['sub_records']['f1_modulate']
print('$lib', 'modulator frac', g['roofline']['frac'], 'kernel_ms', g['roofline']['kernel_ms'], '| headline', l['roofline']['frac'], 'round trip', l['roundtrip_match_rate'])
" | tee -a gpurun_out/r5_exp33.txt
done
