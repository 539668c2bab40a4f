#!/bin/bash
# SQ counter pass (rocprofv3 --pmc, no tracing) over bench.py workloads: where the wave cycles go and what
# the LDS pipe does.   [STREAMS=65536] bash tools/pmc_sq.sh <tag> "<workload> [lib.so]" ...   e.g. "config5" "custom160 tools/libafsk_x.so"
cd "$(dirname "$0")/.."
R=$(pwd); T=$1; shift
export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"
for spec in "$@"; do
  set -- $spec; w=$1; lib=${2:-}
  wl="--workload $w"; case $w in custom*) wl="--workload custom --bauds ${w#custom}" ;; esac
  name=$w; [ -n "$lib" ] && name="${w}_$(basename $lib .so)" && export AFSK_AMD_LIB=$R/$lib || unset AFSK_AMD_LIB
  rm -rf gpurun_out/pmc_sq_$name
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_sq_$name -- python3 $R/bench.py $wl ${STREAMS:+--streams $STREAMS} --sub "" --steps 6 --warmup 2 --preroll-ms 0 --min-region-ms 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-80 )
  python tools/summarize_pmc.py "gpurun_out/pmc_sq_$name" > gpurun_out/${T}_pmc_sq_$name.json
  python -c "import json; d=json.load(open('gpurun_out/${T}_pmc_sq_$name.json')); print('$name', d.get('share_of_wave_cycles'), d.get('share_of_lds_active_cycles'))"
done
