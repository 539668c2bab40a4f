#!/bin/bash
# r6 exp1: what the arbitrary-clock-index path costs with the r5 kernels (before any change).
#   config5 (ci = 0, shortcut) / lead 8 (ci = 8: full search, aligned, hinted) / lead 1 (ci = 1: unaligned, no hint) /
#   random leads (config5_lead)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r6_exp1.txt
: > $O
B="python bench.py --sub '' --steps 20 --warmup 3 --no-cpu-baseline"
for spec in "--workload config5" "--workload custom --bauds 1200 --streams 65536 --lead 8" "--workload custom --bauds 1200 --streams 65536 --lead 1" "--workload config5_lead" "--workload custom --bauds 2400 --streams 65536 --lead random" "--workload custom --bauds 300 --streams 65536 --lead random" "--workload custom --bauds 4000 --streams 65536 --lead random" "--workload custom --bauds 12000 --streams 65536 --lead random"; do
  echo "== $spec" >> $O
  timeout -k 10 300 python bench.py $spec --sub "" --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/r6_exp1.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']
        print(json.dumps({'value': d['value'], 'ms': d['ms_per_step'], 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'alg': r['algorithmic_bytes_per_launch'], 'rt': d.get('roundtrip_match_rate')}))
" >> $O || echo "FAILED" >> $O
done
cat $O
