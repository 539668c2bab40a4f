#!/bin/bash
# bench.py with several builds of the library, one process each, interleaved twice:
#   bash tools/lib_ab.sh "<workload args>" lib1.so lib2.so ...
cd "$(dirname "$0")/.."
args=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    v=$(AFSK_AMD_LIB=$lib timeout 300 python bench.py $args --sub "" --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['full_buffer_gbs'], d['roundtrip_match_rate'])")
    echo "$(basename $lib) $v"
  done
done
