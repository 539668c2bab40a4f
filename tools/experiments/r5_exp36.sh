#!/bin/bash
# staged grouped launch: the new test first, then bench.py's mixed workloads (batch.demod_batch takes the staged form now)
cd "$(dirname "$0")/../.."
rm -f gpurun_out/r5_exp36.txt
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "staged or grouped_dispatch_plan or many_rates or config3" 2>&1 | tail -5 ) | tee -a gpurun_out/r5_exp36.txt
for spec in "375,160,96,1200" "12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240"; do
  for i in 1 2; do
    timeout -k 10 300 python bench.py --workload custom --bauds $spec --streams 65536 --sub "" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$spec'[:24], 'frac', l['roofline']['frac'], 'kernel_ms', l['roofline']['kernel_ms'], 'ms_per_step', l['ms_per_step'], 'entry', l.get('entry'), 'round trip', l['roundtrip_match_rate'])
" | tee -a gpurun_out/r5_exp36.txt
  done
done
timeout -k 10 300 python bench.py --workload custom --bauds 375,160,96,1200 --streams 4096 --sub "" --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('4096 x 4 rates frac', l['roofline']['frac'], 'kernel_ms', l['roofline']['kernel_ms'], 'ms_per_step', l['ms_per_step'])
" | tee -a gpurun_out/r5_exp36.txt
