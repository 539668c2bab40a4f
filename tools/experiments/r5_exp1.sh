#!/bin/bash
# r5 visit 1: the new N > 1 rehearsals (launcher deadline / failure lines, value_no_gather) on the GPU, then the
# ATTRIBUTION pass the r4 verdict asked for before touching the kernels: dynamic instruction counts per wave and
# the wave-cycle split for (a) 375 baud through its uniform kernel vs the same streams through the per-stream
# kernel, (b) the short-symbol rates against 1200 / 2000 baud, (c) a general-piece rate.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
if [ "${1:-a}" = "a" ]; then
( timeout -k 10 1100 python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -8 ) | tee gpurun_out/r5_exp1_pytest_multi.log
for spec in "config5|--workload config5" "custom375|--workload custom --bauds 375 --streams 65536" \
            "custom375mixed|--workload custom --bauds 375 --streams 65536 --entry mixed" \
            "custom6000|--workload custom --bauds 6000 --streams 65536" "custom12000|--workload custom --bauds 12000 --streams 65536" \
            "custom160|--workload custom --bauds 160 --streams 65536"; do
  name=${spec%%|*}; args=${spec#*|}
  timeout -k 10 300 python bench.py $args --steps 20 --warmup 3 --sub "" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['entry'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roundtrip_match_rate'])" | tee -a gpurun_out/r5_exp1_frac.txt
done
else
bash tools/pmc_sets.sh r5a "config5|--workload config5" "u375|--workload custom --bauds 375 --streams 65536" \
   "m375|--workload custom --bauds 375 --streams 65536 --entry mixed" "u6000|--workload custom --bauds 6000 --streams 65536" \
   "u12000|--workload custom --bauds 12000 --streams 65536" "u2000|--workload custom --bauds 2000 --streams 65536" \
   "u3000|--workload custom --bauds 3000 --streams 65536" "u160|--workload custom --bauds 160 --streams 65536" \
   "m4x|--workload custom --bauds 375,160,96,1200 --streams 4096" 2>&1 | tee gpurun_out/r5_exp1_pmc.txt
fi
