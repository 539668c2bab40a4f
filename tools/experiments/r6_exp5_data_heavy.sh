#!/bin/bash
# r6 exp5: data-heavy streams -- training_time 0.05 s instead of the reference's default 0.5 s: almost the whole second is
# data symbols (squelch amplitude per symbol, ECC flushes).  How far from the training-heavy BASELINE shape?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r6_exp5.txt
: > $O
for spec in "--bauds 1200" "--bauds 1200 --training-time 0.05" "--bauds 300 --training-time 0.05" "--bauds 2400 --training-time 0.05" "--bauds 600 --training-time 0.05" "--bauds 4000 --training-time 0.05" "--bauds 12000 --training-time 0.05" "--bauds 800 --training-time 0.05" "--bauds 160 --training-time 0.05"; do
  echo "== $spec ${AFSK_AMD_LIB:-}" >> $O
  timeout -k 10 300 python bench.py --workload custom --streams 65536 $spec --sub "" --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/r6_exp5.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']
        print(json.dumps({'ms': d['ms_per_step'], 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'alg': r['algorithmic_bytes_per_launch'], 'rt': d.get('roundtrip_match_rate')}))
" >> $O || echo "FAILED" >> $O
done
cat $O
