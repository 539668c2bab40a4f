#!/bin/bash
# validation of the r5 kernels (-structurizecfg-skip-uniform-regions build): the whole GPU suite, the long-stream check,
# then the fuzz campaign (36 rates x N random streams x 3 thresholds x both device entries + the all-rate batches)
cd "$(dirname "$0")/../.."
( timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) | tee gpurun_out/r5_validate_pytest.log
( timeout -k 10 400 python tools/long_stream_check.py 2>&1 | tail -3 ) | tee gpurun_out/r5_validate_long.txt
( timeout -k 10 1000 python tools/fuzz_gpu.py ${1:-10000} ${2:-5005} 2>&1 | tail -12 ) | tee gpurun_out/r5_fuzz.txt
