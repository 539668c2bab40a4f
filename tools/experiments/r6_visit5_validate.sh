#!/bin/bash
# r6 visit 5: validation of the re-based kernels.
#  (1) the SAFE code generation (AFSK_SAFE_CODEGEN=1: without -mllvm -structurizecfg-skip-uniform-regions): GPU suite + one fuzz seed
#  (2) the shipped (fast) build: GPU suite + one fuzz seed
#  (3) A/B of the two builds on config5 / config5_lead
#  (4) ragged plans: window sizes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r6_safe_codegen.txt
: > $O
echo "== safe build (tools/libafsk_safe.so = AFSK_SAFE_CODEGEN=1 build.sh): pytest -m gpu" >> $O
( AFSK_AMD_LIB=$PWD/tools/libafsk_safe.so timeout -k 10 1200 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) >> $O
echo "== safe build: tools/fuzz_gpu.py 10000 6601" >> $O
( AFSK_AMD_LIB=$PWD/tools/libafsk_safe.so timeout -k 10 900 python tools/fuzz_gpu.py 10000 6601 2>&1 | tail -8 ) >> $O
cat $O
F=gpurun_out/r6_fast_validate.txt
: > $F
echo "== shipped build: pytest -m gpu" >> $F
( timeout -k 10 1200 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) >> $F
echo "== shipped build: tools/fuzz_gpu.py 10000 6602" >> $F
( timeout -k 10 900 python tools/fuzz_gpu.py 10000 6602 2>&1 | tail -8 ) >> $F
cat $F
echo "== A/B safe vs fast" >> $O
( timeout -k 10 600 python tools/lib_ab.py --bauds 1200 --streams 65536 afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_safe.so 2>&1 | tail -12 ) >> $O
( timeout -k 10 600 python tools/lib_ab.py --bauds 12000 --streams 65536 afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_safe.so 2>&1 | tail -6 ) >> $O
( timeout -k 10 600 python tools/lib_ab.py --bauds 300,1200,2400,4000 --streams 65536 afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_safe.so 2>&1 | tail -6 ) >> $O
R=gpurun_out/r6_ragged_windows.txt
: > $R
for w in 4096 2048 8192 16384 0; do
  echo "== AFSK_GROUP_WINDOW=$w" >> $R
  AFSK_GROUP_WINDOW=$w timeout -k 10 300 python bench.py --sub ragged_lengths --steps 3 --warmup 1 --no-cpu-baseline --next-reps 10 --preroll-ms 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        print(json.dumps(json.loads(l)['sub_records']['ragged_lengths']))
" >> $R
done
cat $R
