#!/bin/bash
# margins test hoisted out of the slice loop + cheaper mark/space combination (multi_rounds): library before / after
cd "$(dirname "$0")/../.."
R=$(pwd)
for rep in 1 2; do
for b in 6000 12000 4000 3000 2000; do
  for n in 65536 4096; do
    for lib in tools/libafsk_prev.so afskmodem_amd/csrc/libafsk_amd.so; do
      AFSK_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py --workload custom --bauds $b --streams $n --steps $((n > 10000 ? 30 : 200)) --sub "" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b x $n $(basename $lib)', d['roofline']['frac'], d['roofline']['kernel_ms'], d['roundtrip_match_rate'])"
    done
  done
done
done
