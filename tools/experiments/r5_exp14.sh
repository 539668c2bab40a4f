#!/bin/bash
# k12: probes closer than a round + partial rounds at the hint boundary, against k11
cd "$(dirname "$0")/../.."
T=tools
rm -f gpurun_out/r5_exp14.txt
for spec in "--bauds 4000" "--bauds 3000" "--bauds 12000" "--bauds 6000" "--bauds 1200" "--bauds 160" "--bauds 96" "--bauds 375" "--bauds 800" "--bauds 2400" "--bauds 300" "--bauds 300,1200,2400" "--bauds 375,160,96,1200" "--bauds 4000 --streams 4096 --reps 40"; do
  timeout -k 10 300 python tools/lib_ab.py $spec --rounds 8 $T/libafsk_k11.so $T/libafsk_k12.so 2>&1 | grep -v "^bench.py\|Warning\|warn\|amdgpu.ids" | tee -a gpurun_out/r5_exp14.txt
done
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 ) | tee gpurun_out/r5_exp14_pytest.log
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/gpurun_out/pmc_fetch_k12_4000 -- python3 $OLDPWD/bench.py --workload custom --bauds 4000 --streams 65536 --sub "" --steps 6 --warmup 2 --preroll-ms 0 --min-region-ms 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-80 )
python tools/summarize_pmc.py gpurun_out/pmc_fetch_k12_4000 | grep -A3 FETCH_SIZE | head -5
