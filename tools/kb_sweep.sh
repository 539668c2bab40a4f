#!/bin/bash
# kbench over a list of baud rates, current library against KBENCH_LIB_B (default: the round-1 build
# tools/libafsk_r1.so when present).   bash tools/kb_sweep.sh "800 500 480 400 1200" [streams]
cd "$(dirname "$0")"
export KBENCH_LIB_B=${KBENCH_LIB_B:-$(pwd)/libafsk_r1.so}
for b in $1; do
  echo "=== baud $b"
  timeout 300 ./kbench ${2:-4096} $b 12 5 2>&1 | grep -E "base:|!=|median|HIP error"
done
