#!/bin/bash
# the r5 GPU visit that produces the committed evidence: default bench line, then the rocprofv3 passes (kernel trace +
# FETCH_SIZE + WRITE_SIZE, one run each) for the headline, configs 2 / 3, the slowest rates and the mixed-rate cases
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp PROF_TAG=r5
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) | tee gpurun_out/r5_smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r5_bench_main.err | grep '^{"metric"' ) > gpurun_out/r5_bench_main.json
cut -c1-400 gpurun_out/r5_bench_main.json; cp gpurun_out/bench_full_n1.json gpurun_out/r5_bench_full_n1.json
SKIP_TESTS=1 PROF_ONLY=1 PROF_WL="config5 config2 config3 custom4000 custom3000 custom12000 custom375,160,96,1200" bash tools/gpu_round.sh prof r5 2>&1 | tail -60
# LDS bank-conflict share and instruction counts of the final kernels (375 baud: conflict-free chunk order; 160 / 96 baud: general pieces)
SETS="A B C" bash tools/pmc_sets.sh r5final "config5|--workload config5" "u12000|--workload custom --bauds 12000 --streams 65536" "u375|--workload custom --bauds 375 --streams 65536" \
   "u160|--workload custom --bauds 160 --streams 65536" "u96|--workload custom --bauds 96 --streams 65536" 2>&1 | grep -v "^W2026" | tee gpurun_out/r5_final_pmc.txt
# the same default bench line once more, now that this visit's PMC passes have measured the traffic of THIS kernel source
# (roofline.traffic is only attached when profiles/traffic_latest.json carries the hash of the source that is running)
cp gpurun_out/traffic_latest.json profiles/traffic_latest.json
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r5_bench_main2.err | grep '^{"metric"' ) > gpurun_out/r5_bench_main2.json
cut -c1-300 gpurun_out/r5_bench_main2.json; cp gpurun_out/bench_full_n1.json gpurun_out/r5_bench_full_n1_2.json

