#!/bin/bash
# phase A early-out (total(0) < N: the stream starts with the training sequence, clock index 0 without a search) against the same build without it
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp20.txt
for spec in "--bauds 1200" "--bauds 300" "--bauds 6000" "--bauds 12000" "--bauds 2400" "--bauds 160" "--bauds 800" "--bauds 1200 --streams 4096 --reps 40" "--bauds 300,1200,2400" "--bauds 375,160,96,1200"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 $T/libafsk_k18s.so $T/libafsk_k19.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp20.txt
done
timeout -k 10 300 python bench.py --workload config4 --sub "" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-700 | tee -a gpurun_out/r5_exp20.txt
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 ) | tee gpurun_out/r5_exp20_pytest.log
