#!/bin/bash
# r6 visit 6: the whole GPU suite on the shipped build; the kernel-facing files on the AFSK_SAFE_CODEGEN build
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
F=gpurun_out/r6_v6_suite_fast.txt
( timeout -k 10 1100 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > $F
cat $F | tail -6
S=gpurun_out/r6_v6_suite_safe.txt
( AFSK_AMD_LIB=$PWD/tools/libafsk_safe.so timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_oracle.py tests/test_gpu_rounds.py tests/test_gpu_lead.py tests/test_gpu_next_rows.py -q -m gpu 2>&1 | tail -25 ) > $S
cat $S | tail -6
