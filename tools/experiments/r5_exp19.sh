#!/bin/bash
# k18: bit-sliced two-codeword Hamming decode in the flush + tail hint for every geometry of the per-stream kernel, against k16
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp19.txt
for spec in "--bauds 12000" "--bauds 6000" "--bauds 4000" "--bauds 2400" "--bauds 1200" "--bauds 12000 --streams 8192 --reps 20" "--bauds 300,1200,2400" "--bauds 12000,6000,1200,300"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 10 $T/libafsk_k16.so $T/libafsk_k18.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp19.txt
done
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 ) | tee gpurun_out/r5_exp19_pytest.log
