#!/bin/bash
# r6 exp3: fixed leads 0..8 at 1200 baud, 65536 streams: which shifts (2 * lead) & 15 are slow after the re-base?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r6_exp3.txt
: > $O
for lead in ${LEADS:-0 1 2 3 4 5 6 7 8}; do
  echo "== lead $lead ${AFSK_AMD_LIB:-}" >> $O
  timeout -k 10 300 python bench.py --workload custom --bauds ${BAUD:-1200} --streams 65536 --lead $lead --sub "" --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/r6_exp3.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']
        print(json.dumps({'ms': d['ms_per_step'], 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'rt': d.get('roundtrip_match_rate')}))
" >> $O || echo "FAILED" >> $O
done
cat $O
