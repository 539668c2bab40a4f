#!/bin/bash
# CPU-only sanitizer run of the oracle (GPU ASan is not available on the pool): builds
# oracle/libafsk_oracle_asan.so with -fsanitize=address,undefined and drives every golden
# decode / listen case plus a threaded batch through it.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle libafsk_oracle_asan.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python tools/oracle_asan_run.py
