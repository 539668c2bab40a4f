#!/bin/bash
# Builds afskmodem_amd/csrc/libafsk_amd.so for gfx950 (cross-compiles without a GPU).  The demod
# kernels dominate the build: the two mixed-baud instantiations and one translation unit per uniform
# bit_frames value (afsk_demod_uniform.hip with -DAFSK_UNIFORM_BF=N) compile in parallel.
set -euo pipefail
cd "$(dirname "$0")"
ARCH=${AFSK_ARCH:-gfx950}
JOBS=${AFSK_BUILD_JOBS:-$(nproc)}
# -structurizecfg-skip-uniform-regions (r5): the AMDGPU backend structurizes wave-UNIFORM branches too by default
# (flag registers, s_mov_b64 / s_andn2_b64 / extra s_cbranch per if), and these kernels are full of them -- one wave
# = one stream, the whole receiver state machine is scalar.  A wave issues one instruction every four cycles
# whatever its kind, so the scalar bookkeeping costs as much as the vector arithmetic: leaving uniform regions
# as plain branches removes ~10 % of the static scalar instructions and 2 - 6 % of the kernel time
# (profiles/r5_exp4_lib_ab.txt).  The flag is an LLVM-internal one: the bit-exact GPU suite and tools/fuzz_gpu.py
# are what vouch for the code it produces.
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=${ARCH} -Wall -Wno-unused-function -mllvm -structurizecfg-skip-uniform-regions"
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
# bit_frames values with a compile-time geometry = AFSK_FAST_BF_LIST + AFSK_GP_BF_LIST in afsk_demod_impl.h; 0 = run-time geometry
UNIFORM_BF=$(sed -n 's/^#define AFSK_\(FAST\|GP\)_BF_LIST(X)//p' afsk_demod_impl.h | tr -d 'X()' | tr '\n' ' ')
WANT=$(sed -n 's/^static_assert(kUniformBfCount == \([0-9]*\),.*/\1/p' afsk_demod_impl.h)
HAVE=$(echo $UNIFORM_BF | wc -w)
if [ -z "$WANT" ] || [ "$HAVE" -ne "$WANT" ]; then
  echo "build.sh: scraped $HAVE bit_frames values from afsk_demod_impl.h, the header says ${WANT:-?} (kUniformBfCount): the AFSK_*_BF_LIST macros must stay on one line each" >&2
  exit 1
fi
{
  # longest jobs first
  for f in afsk_demod_small afsk_demod_big; do echo "$f.o $f.hip"; done
  for b in 0 $(for b in $UNIFORM_BF; do echo $b; done | sort -rn); do echo "afsk_demod_uniform_$b.o afsk_demod_uniform.hip -DAFSK_UNIFORM_BF=$b"; done
  for f in afsk_capi afsk_demod afsk_synth afsk_gate; do echo "$f.o $f.hip"; done
} > "$OBJ/jobs"
EXTRA="$*"
export FLAGS OBJ EXTRA
xargs -P "$JOBS" -L 1 bash -c 'hipcc $FLAGS -c -o "$OBJ/$0" "$@" $EXTRA' < "$OBJ/jobs"
hipcc ${FLAGS} -shared -o libafsk_amd.so "$OBJ"/*.o
echo "built $(pwd)/libafsk_amd.so"
