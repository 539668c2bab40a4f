#!/bin/bash
# round 4 experiment 2: mechanics of the grouped dispatch (AFSK_GROUP_MODE 0 split / 1 split, all on side streams /
# 2 serial / 3 fused) against the per-stream kernel, and rate order (cycle vs blocks) for the per-stream kernel
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
R=$(pwd)
one() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
f = json.load(open('$R/' + d['full_record']))
print('  frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'host_issue_ms', f['host_issue_ms_per_step'], d['entry'], d['roundtrip_match_rate'])"; }
R18=12000,375,250,240,160,125,120,96,80,75,60,50,48,40,32,30,25,24
for wl in "--workload config3" "--workload custom --bauds 375,160,96,1200" "--workload custom --bauds 375,160,96,1200 --streams 16384" "--workload custom --bauds 375,160,96,1200 --streams 65536 --steps 20" "--workload custom --bauds $R18 --streams 65536 --steps 20" "--workload custom --bauds $R18 --streams 4096"; do
  echo "== $wl"
  echo " mixed, cycle"; timeout -k 10 300 python bench.py $wl --entry mixed --sub "" --no-cpu-baseline 2>/dev/null | one
  echo " mixed, blocks"; timeout -k 10 300 python bench.py $wl --entry mixed --rate-order blocks --sub "" --no-cpu-baseline 2>/dev/null | one
  for m in 0 1 2 3; do
    echo " grouped mode $m"; AFSK_GROUP_MODE=$m timeout -k 10 300 python bench.py $wl --sub "" --no-cpu-baseline 2>/dev/null | one
  done
done
