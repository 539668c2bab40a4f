#!/bin/bash
# hipcc's own resource remark (-Rpass-analysis=kernel-resource-usage) for every shipped demod kernel, one
# line each (cross-compiles gfx950 without a GPU).   bash tools/kernel_resources.sh > profiles/rN_kernel_resources.txt
cd "$(dirname "$0")/../afskmodem_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -structurizecfg-skip-uniform-regions -Rpass-analysis=kernel-resource-usage -c -o /dev/null"
UNIFORM_BF=$(sed -n 's/^#define AFSK_\(FAST\|GP\)_BF_LIST(X)//p' afsk_demod_impl.h | tr -d 'X()' | tr '\n' ' ')
WANT=$(sed -n 's/^static_assert(kUniformBfCount == \([0-9]*\),.*/\1/p' afsk_demod_impl.h)
[ "$(echo $UNIFORM_BF | wc -w)" -eq "${WANT:-0}" ] || { echo "kernel_resources.sh: bit_frames lists do not match kUniformBfCount" >&2; exit 1; }
{
  echo "afsk_demod_small.hip"; echo "afsk_demod_big.hip"
  for b in 0 $UNIFORM_BF; do echo "afsk_demod_uniform.hip -DAFSK_UNIFORM_BF=$b"; done
} | FLAGS="$FLAGS" xargs -P "${JOBS:-8}" -L 1 bash -c 'hipcc $FLAGS "$0" $1 2>&1 | python3 -c "
import re, sys
cur = None
for ln in sys.stdin:
    m = re.search(r\"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)\", ln)
    if not m: continue
    if m.group(1) == \"Function Name\":
        if cur: print(cur)
        n = m.group(2)
        u = re.search(r\"demod_uniform_kernel_tILi(\d+)ELi0ELb([01])\", n)
        k = re.search(r\"demod_kernel_tILi0ELi4ELi0ELb([01])\", n)
        name = (\"uniform bf=%4s %s\" % (u.group(1) if u.group(1) != \"0\" else \"rt\", \"large\" if u.group(2) == \"1\" else \"small\")) if u else ((\"mixed          %s\" % (\"large\" if k.group(1) == \"1\" else \"small\")) if k else n)
        cur = name + \":\"
    else:
        short = {\"TotalSGPRs\": \"SGPR\", \"VGPRs\": \"VGPR\", \"AGPRs\": \"AGPR\", \"ScratchSize [bytes/lane]\": \"scratch\", \"Occupancy [waves/SIMD]\": \"waves/SIMD\", \"SGPRs Spill\": \"SGPR-spill\", \"VGPRs Spill\": \"VGPR-spill\", \"LDS Size [bytes/block]\": \"LDS/block\"}[m.group(1)]
        cur += \" %s %s\" % (short, m.group(2))
if cur: print(cur)
"' | sort -k1,1 -k2,2 -V
