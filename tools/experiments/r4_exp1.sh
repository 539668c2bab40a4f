#!/bin/bash
# round 4 experiments: grouped-dispatch overlap, all-geometry mixed kernel, ingest breakdown
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
echo "== anyorder probe"; for a in "30 4 256" "30 4 512" "200 3 4096"; do timeout -k 5 60 tools/anyorder_probe $a; done
echo "== pytest (new tests)"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grouped_dispatch_plan or full_size_config4 or device_side or wav or gate" 2>&1 | tail -4
one() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
f = json.load(open('$R/' + d['full_record']))
print({k: d[k] for k in ('value', 'ms_per_step', 'entry', 'roundtrip_match_rate')}, 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'host_issue_ms', f['host_issue_ms_per_step'])"; }
for lib in afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_mixedall.so; do
  for args in "--workload config3 --entry mixed" "--workload custom --bauds 375,160,96,1200 --entry mixed" "--workload custom --bauds 375,160,96,1200 --streams 65536 --entry mixed --steps 20" "--workload custom --bauds 12000,375,250,240,160,125,120,96,80,75,60,50,48,40,32,30,25,24 --streams 65536 --entry mixed --steps 20"; do
    echo "== $(basename $lib) $args"
    AFSK_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py $args --sub "" --no-cpu-baseline 2>/dev/null | one
  done
done
echo "== grouped, 18 rates x 65536"
timeout -k 10 300 python bench.py --workload custom --bauds 12000,375,250,240,160,125,120,96,80,75,60,50,48,40,32,30,25,24 --streams 65536 --steps 20 --sub "" --no-cpu-baseline 2>/dev/null | one
echo "== timeline of the grouped dispatch (4 rates x 1024 streams)"
rm -rf gpurun_out/prof_grouped
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_grouped -- python3 $R/bench.py --workload custom --bauds 375,160,96,1200 --sub "" --steps 30 --warmup 3 --preroll-ms 0 --min-region-ms 0 --no-cpu-baseline > /dev/null 2>&1 )
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_grouped/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'demod' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[-40]['Start_Timestamp'])
for r in rows[-40:-16]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} .. {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us  q={r.get('Queue_Id')} grid={r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size')} {r['Kernel_Name'][:60]}")
PY
echo "== ingest breakdown"
for env in "AFSK_IO_THREADS=32 AFSK_INGEST_SLOTS=4" "AFSK_IO_THREADS=32 AFSK_INGEST_SLOTS=8" "AFSK_IO_THREADS=64 AFSK_INGEST_SLOTS=8" "AFSK_IO_THREADS=128 AFSK_INGEST_SLOTS=8" "AFSK_IO_THREADS=64 AFSK_INGEST_SLOTS=8 AFSK_INGEST_WINDOW_MB=8" "AFSK_IO_THREADS=64 AFSK_INGEST_SLOTS=4 AFSK_INGEST_WINDOW_MB=32" "AFSK_IO_THREADS=16 AFSK_INGEST_SLOTS=8"; do
  echo "-- $env"
  env $env AFSK_INGEST_STATS=1 timeout -k 10 300 python tools/wav_ingest_bench.py --reps 5 2> gpurun_out/ingest_stats.err | python -c "
import sys, json
d = json.load(sys.stdin)
print('load_wav_batch', d['native_ingest'], 'pinned', d['pinned_hipMemcpy_same_bytes'], 'e2e', d['load_batch_end_to_end'])"
  grep "afsk_wav_ingest:" gpurun_out/ingest_stats.err | sort -t, -k3 | sed -n '3,4p'
done
