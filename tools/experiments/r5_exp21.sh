#!/bin/bash
# phase C: "every symbol of the round / pass loud" pre-test (largest quiet sum per lane; zero test before the
# ballot compaction) against the same build without it
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp21.txt
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 ) | tee gpurun_out/r5_exp21_pytest.log
for spec in "--bauds 12000" "--bauds 6000" "--bauds 4000" "--bauds 3000" "--bauds 2000" "--bauds 1500" "--bauds 1000" "--bauds 750" "--bauds 2400" "--bauds 1200" "--bauds 600" "--bauds 300" "--bauds 160" "--bauds 40" "--bauds 1200 --streams 4096 --reps 40" "--bauds 300,1200,2400" "--bauds 375,160,96,1200" "--bauds 375,160,96,1200 --streams 4096 --reps 40"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 $T/libafsk_k19.so $T/libafsk_k20.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp21.txt
done
