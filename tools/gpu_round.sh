#!/bin/bash
# One GPU visit: parity tests, smoke, the bench line (N = 1: config5 headline + sub-records config2/3/4
# + f1/f2/f3), and with "prof" rocprofv3 kernel trace + PMC passes (FETCH_SIZE and WRITE_SIZE in
# separate runs) for the workloads listed in PROF_WL.  Outputs under gpurun_out/.
#   [SKIP_TESTS=1] [PROF_WL="config5 config2 custom375,160"] bash tools/gpu_round.sh [prof] [tag]
# With "prof" the LAST action is the check that gpurun_out/traffic_latest.json -- what bench.py reports as
# roofline.traffic -- was measured on the kernel sources of this tree (tests/test_evidence_current.py checks the
# committed copy on the CPU): run this as the last GPU visit of a round, then tools/collect_profiles.sh <tag>.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROF_TAG=${2:-r6}
T=$PROF_TAG
if [ -z "${SKIP_TESTS:-}" ]; then
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) | tee gpurun_out/${T}_pytest_gpu.log
fi
# The validated escape hatch stays validated: when tools/libafsk_safe.so exists (AFSK_SAFE_CODEGEN=1 AFSK_OUT=$PWD/tools/libafsk_safe.so
# bash afskmodem_amd/csrc/build.sh -- the build WITHOUT the LLVM-internal code-generation flag) the kernel-facing test
# files run against it too, and both builds decode one resident batch in one process with their outputs compared
if [ -z "${SKIP_TESTS:-}" ] && [ -f tools/libafsk_safe.so ] && [ tools/libafsk_safe.so -nt afskmodem_amd/csrc/afsk_demod_ring.h ]; then
  ( AFSK_AMD_LIB=$PWD/tools/libafsk_safe.so timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_oracle.py tests/test_gpu_rounds.py tests/test_gpu_lead.py -q -m gpu 2>&1 | tail -3 ) | tee gpurun_out/${T}_pytest_gpu_safe_codegen.log
  ( timeout 300 python tools/lib_ab.py --bauds 300,1200,2400,4000 --lead random afskmodem_amd/csrc/libafsk_amd.so tools/libafsk_safe.so 2>&1 | tail -3 ) | tee -a gpurun_out/${T}_pytest_gpu_safe_codegen.log
fi
if [ -z "${PROF_ONLY:-}" ]; then
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) | tee gpurun_out/${T}_smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench_main.err | grep '^{"metric"' ) > gpurun_out/${T}_bench_main.json
cut -c1-600 gpurun_out/${T}_bench_main.json
fi
if [ "${1:-}" = "prof" ]; then
  R=$(pwd)
  for w in ${PROF_WL:-config5 config5_lead config2 config3}; do
    steps=200; wl="--workload $w"; kern=demod
    case $w in
      config5|config3|config4|config5_lead) steps=20 ;;
      custom*) wl="--workload custom --bauds ${w#custom} --streams ${PROF_STREAMS:-65536}"; steps=20 ;;
    esac
    rm -rf gpurun_out/prof_trace_$w gpurun_out/prof_pmc1_$w gpurun_out/prof_pmc2_$w
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace_$w -- python3 $R/bench.py $wl --sub "" --steps $steps --warmup 3 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tee $R/gpurun_out/prof_bench_$w.json | cut -c1-300 )
    ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_pmc1_$w -- python3 $R/bench.py $wl --sub "" --steps 6 --warmup 2 --preroll-ms 0 --min-region-ms 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100 )
    ( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_pmc2_$w -- python3 $R/bench.py $wl --sub "" --steps 6 --warmup 2 --preroll-ms 0 --min-region-ms 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100 )
    python tools/summarize_prof.py gpurun_out $w $steps 2>&1 | tail -40
    python tools/trace_timeline.py gpurun_out/prof_trace_$w 2>&1 | tee gpurun_out/timeline_$w.txt
  done
  # the SURVEY 8(f) rows f1 / f2: one run of the config5 headline carrying both sub-records
  w=next
  rm -rf gpurun_out/prof_trace_$w gpurun_out/prof_pmc1_$w gpurun_out/prof_pmc2_$w
  NX="--sub f1_modulate,f2_gate --steps 3 --warmup 1 --preroll-ms 0 --min-region-ms 0 --no-cpu-baseline --next-reps 20"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace_$w -- python3 $R/bench.py $NX 2>&1 | grep '^{"metric"' | tee $R/gpurun_out/prof_bench_$w.json | cut -c1-200 )
  ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_pmc1_$w -- python3 $R/bench.py $NX --next-reps 4 2>&1 | tail -1 | cut -c1-100 )
  ( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_pmc2_$w -- python3 $R/bench.py $NX --next-reps 4 2>&1 | tail -1 | cut -c1-100 )
  python tools/summarize_prof.py gpurun_out $w 20 --kernel modulate_kernel --name modulate 2>&1 | tail -30
  python tools/summarize_prof.py gpurun_out $w 20 --kernel block_amp_kernel --name gate 2>&1 | tail -30
  python tools/summarize_prof.py gpurun_out $w 20 --kernel gate_scan_kernel --name gate_scan 2>&1 | tail -12
  # ---- last: the traffic file of this visit belongs to the kernel sources of this tree, and covers the headline
  python - <<'PY' || { echo "gpu_round.sh: gpurun_out/traffic_latest.json is NOT current -- do not commit it" >&2; exit 1; }
import json, sys
sys.path.insert(0, ".")
import bench
tj = json.load(open("gpurun_out/traffic_latest.json"))
h = bench.kernel_source_hash()
assert tj["kernel_source_hash"] == h, (tj["kernel_source_hash"], h)
for w in ("config5", "config5_lead", "config2", "config3"):
    assert w in tj["entries"], w
print("traffic_latest.json: kernel source", h, {k: v["hbm_bytes_per_launch"] for k, v in tj["entries"].items()})
PY
fi
