#!/bin/bash
# round 4 experiment 3: uniform-kernel variants for the slowest rates (steady state = 65536 streams), + the box's CPU share
cd "$(dirname "$0")/../.."
R=$(pwd)
echo "== cpu share"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; cat /proc/self/cgroup | head -3
one() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], d['roundtrip_match_rate'])"; }
for rep in 1 2; do
for lib in afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_wm128.so tools/libafsk_hintall.so; do
  for wl in "--bauds 375" "--bauds 12000" "--bauds 6000" ; do
    case "$lib $wl" in *wm128*12000*|*wm128*6000*|*hintall*375*) continue;; esac
    echo "== $(basename $lib) $wl x65536"; AFSK_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py --workload custom $wl --streams 65536 --steps 30 --sub "" --no-cpu-baseline 2>/dev/null | one
    echo "== $(basename $lib) $wl x4096"; AFSK_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py --workload custom $wl --streams 4096 --steps 200 --sub "" --no-cpu-baseline 2>/dev/null | one
  done
done
done
echo "== hintall, mixed kernel"
for lib in afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_hintall.so; do
  echo "-- $(basename $lib) config3"; AFSK_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py --workload config3 --sub "" --no-cpu-baseline 2>/dev/null | one
  echo "-- $(basename $lib) 12000,6000,1200"; AFSK_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py --workload custom --bauds 12000,6000,1200 --streams 65536 --steps 30 --sub "" --no-cpu-baseline 2>/dev/null | one
done
echo "== parity of the variants (uniform + mixed entries, quick)"
for lib in tools/libafsk_wm128.so tools/libafsk_hintall.so; do
  AFSK_AMD_LIB=$R/$lib timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_cases_device or large_launch or runtime_geometry_rates or every_alignment or fuzz_noise" 2>&1 | tail -2
done
