#!/bin/bash
# mixed-rate evidence: four rates at 4096 streams, eighteen rates at 65536 streams (one rocprofv3 run per figure)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp PROF_TAG=r5
SKIP_TESTS=1 PROF_ONLY=1 PROF_STREAMS=4096 PROF_WL="custom375,160,96,1200" bash tools/gpu_round.sh prof r5 2>&1 | grep -v "^W2026" | tail -5
mv gpurun_out/prof_summary_custom375,160,96,1200.json gpurun_out/prof_summary_mix4_4096.json
mv gpurun_out/prof_bench_custom375,160,96,1200.json gpurun_out/prof_bench_mix4_4096.json
SKIP_TESTS=1 PROF_ONLY=1 PROF_WL="custom12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240" bash tools/gpu_round.sh prof r5 2>&1 | grep -v "^W2026" | tail -5
mv "gpurun_out/prof_summary_custom12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240.json" gpurun_out/prof_summary_mix18_65536.json
mv "gpurun_out/prof_bench_custom12000,6000,4000,3000,2400,2000,1500,1200,1000,800,750,600,500,480,400,375,300,240.json" gpurun_out/prof_bench_mix18_65536.json
( timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -q -m gpu -x -k "torchrun or failure_paths" 2>&1 | tail -4 ) | tee gpurun_out/r5_round_b_pytest.log
