#!/bin/bash
# Builds afskmodem_amd/csrc/libafsk_amd.so for gfx950 (cross-compiles without a GPU).  The demod
# kernels dominate the build: the two mixed-baud instantiations and one translation unit per uniform
# bit_frames value (afsk_demod_uniform.hip with -DAFSK_UNIFORM_BF=N) compile in parallel.
set -euo pipefail
cd "$(dirname "$0")"
ARCH=${AFSK_ARCH:-gfx950}
JOBS=${AFSK_BUILD_JOBS:-$(nproc)}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=${ARCH} -Wall -Wno-unused-function"
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
