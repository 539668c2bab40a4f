#!/bin/bash
# rocprofv3 --pmc passes (no tracing) with several counter SETS over bench.py workloads: where the wave cycles go
# (set A), how many instructions of each kind a wave executes (set B), instruction cache and LDS conflicts (set C).
#   [STREAMS=65536] [SETS="A B C"] bash tools/pmc_sets.sh <tag> "<name>|<bench.py args>" ...
#   e.g.  bash tools/pmc_sets.sh r5 "u375|--workload custom --bauds 375" "m375|--workload custom --bauds 375 --entry mixed"
# -> gpurun_out/<tag>_pmc_<name>.json  (per-kernel averages of every counter + derived shares)
cd "$(dirname "$0")/.."
R=$(pwd); T=$1; shift
export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_IFETCH SQ_WAVES"
Cc="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_UNALIGNED_STALL"
for spec in "$@"; do
  name=${spec%%|*}; args=${spec#*|}
  dirs=""
  for set in ${SETS:-A B C}; do
    case $set in A) C="$A" ;; B) C="$B" ;; *) C="$Cc" ;; esac
    d=gpurun_out/pmc_${T}_${name}_$set
    rm -rf $d
    ( cd /tmp && timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $R/$d -- python3 $R/bench.py $args ${STREAMS:+--streams $STREAMS} --sub "" --steps 6 --warmup 2 --preroll-ms 0 --min-region-ms 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-60 ) || exit 1
    dirs="$dirs $d"
  done
  python tools/summarize_pmc.py "$dirs" > gpurun_out/${T}_pmc_$name.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${T}_pmc_$name.json"))
print("$name", json.dumps(d.get("per_wave")), json.dumps(d.get("share_of_wave_cycles")), json.dumps(d.get("share_of_lds_active_cycles")), json.dumps(d.get("icache")))
PY
done
