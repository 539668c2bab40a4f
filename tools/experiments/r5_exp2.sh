#!/bin/bash
# r5 kernel variants against the r4 library, both loaded into ONE kbench process (interleaved launches, outputs
# compared with the independent two-pass baseline): bash tools/experiments/r5_exp2.sh "<libs, colon separated>" "<bauds>" [streams]
cd "$(dirname "$0")/.."   # (tools/: kbench lives there)
export KBENCH_LIB_B=$1 KBENCH_NO_MIXED=1
for n in ${3:-65536}; do
for b in $2; do
  echo "=== baud $b streams $n"
  timeout -k 10 300 ./kbench $n $b 14 5 2>&1 | grep -E "base:|!=|median|HIP error"
done
done
