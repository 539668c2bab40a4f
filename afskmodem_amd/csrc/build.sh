#!/bin/bash
# Builds afskmodem_amd/csrc/libafsk_amd.so for gfx950 (cross-compiles without a GPU).  The demod
# kernels dominate the build: the two mixed-baud instantiations and one translation unit per uniform
# bit_frames value (afsk_demod_uniform.hip with -DAFSK_UNIFORM_BF=N) compile in parallel.
#
#   AFSK_SAFE_CODEGEN=1    build WITHOUT the LLVM-internal code-generation flag below (the validated escape hatch
#                          for a toolchain bump: profiles/r6_safe_codegen.txt has the GPU suite + fuzz on that build)
#   AFSK_FAST_CODEGEN=1    keep the flag although the compiler is not the one it was validated with
#   AFSK_OUT=<file.so>     output name (default libafsk_amd.so; tests/test_isa_lint.py and A/B builds)
#   AFSK_KEEP_ASM=<dir>    also leave the device assembly (*.s) of every demod translation unit there
#   AFSK_BUILD_JOBS=<n>    parallel compiles (default: nproc)
#   AFSK_PRINT_FLAGS=1     print the compile flags this invocation would use and exit (tests/test_isa_lint.py)
#   AFSK_VALIDATED_TOOLCHAIN=<file>  another VALIDATED_TOOLCHAIN file (the same test: a compiler that is not the validated one)
set -euo pipefail
cd "$(dirname "$0")"
# The kernels are written for gfx950 alone: hand-placed wait states around inline asm (VALU-written SGPRs read by
# v_writelane_b32 / produced by v_cmp_*_sdwa: afsk_demod_phasec.h, afsk_demod_rounds_multi.h) follow gfx940 / gfx950's
# hazard rules, SDWA with an SGPR destination does not exist from gfx10 on, LDS-DMA and DPP row_bcast are gfx9.
ARCH=${AFSK_ARCH:-gfx950}
if [ "$ARCH" != "gfx950" ]; then
  echo "build.sh: AFSK_ARCH=$ARCH refused -- these kernels are gfx950 (MI355X) code: inline asm with that target's hazard wait states, LDS-DMA, DPP row_bcast" >&2
  exit 1
fi
JOBS=${AFSK_BUILD_JOBS:-$(nproc)}
OUT=${AFSK_OUT:-libafsk_amd.so}
# -structurizecfg-skip-uniform-regions (r5): the AMDGPU backend structurizes wave-UNIFORM branches too by default
# (flag registers, s_mov_b64 / s_andn2_b64 / extra s_cbranch per if), and these kernels are full of them -- one wave
# = one stream, the whole receiver state machine is scalar.  A wave issues one instruction every four cycles
# whatever its kind, so the scalar bookkeeping costs as much as the vector arithmetic: leaving uniform regions
# as plain branches removes ~10 % of the static scalar instructions and 2 - 6 % of the kernel time
# (profiles/r5_exp4_lib_ab.txt).  The flag is an LLVM-internal one: the bit-exact GPU suite and tools/fuzz_gpu.py
# are what vouch for the code it produces -- for ONE compiler.  VALIDATED_TOOLCHAIN names it (the `hipcc --version`
# lines the suite + fuzz campaign last ran against); with any other compiler the build falls back to the default
# code generation (AFSK_SAFE_CODEGEN: validated too, a few per cent slower) unless AFSK_FAST_CODEGEN=1 insists.
# tests/test_isa_lint.py checks, on the CPU, what of the generated code can be checked without a GPU.
FAST_FLAG="-mllvm -structurizecfg-skip-uniform-regions"
HAVE_TC=$(hipcc --version 2>/dev/null | grep -E '^(HIP version|AMD clang version)' | sed 's/ *$//' || true)
WANT_TC=$(grep -v '^#' "${AFSK_VALIDATED_TOOLCHAIN:-VALIDATED_TOOLCHAIN}" 2>/dev/null | sed 's/ *$//' || true)
CODEGEN=fast
if [ -n "${AFSK_SAFE_CODEGEN:-}" ] && [ "${AFSK_SAFE_CODEGEN}" != "0" ]; then
  CODEGEN=safe
elif [ "$HAVE_TC" != "$WANT_TC" ] && [ -z "${AFSK_FAST_CODEGEN:-}" ]; then
  echo "build.sh: WARNING: this compiler is not the one the uniform-region flag was validated with:" >&2
  echo "$HAVE_TC" | sed 's/^/    have: /' >&2
  echo "$WANT_TC" | sed 's/^/    want: /' >&2
  echo "  building with the default code generation (AFSK_SAFE_CODEGEN); after the GPU suite + tools/fuzz_gpu.py pass with" >&2
  echo "  AFSK_FAST_CODEGEN=1, update afskmodem_amd/csrc/VALIDATED_TOOLCHAIN" >&2
  CODEGEN=safe
fi
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=${ARCH} -Wall -Wno-unused-function"
LINKFLAGS="$FLAGS"
[ "$CODEGEN" = fast ] && FLAGS="$FLAGS $FAST_FLAG"
if [ -n "${AFSK_PRINT_FLAGS:-}" ]; then echo "$CODEGEN: $FLAGS"; exit 0; fi
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
SRC=$(pwd)
ASM=${AFSK_KEEP_ASM:-}
[ -n "$ASM" ] && mkdir -p "$ASM" && ASM=$(cd "$ASM" && pwd)
export FLAGS OBJ EXTRA SRC ASM
# (with AFSK_KEEP_ASM every compile runs in its own directory: -save-temps names its files after the source, and the
# uniform translation units share one)
xargs -P "$JOBS" -L 1 bash -c '
  o=$0; src=$1; shift
  if [ -n "$ASM" ] && [[ "$o" == afsk_demod_* ]] && [ "$o" != afsk_demod.o ]; then
    d="$OBJ/tmp_${o%.o}"; mkdir -p "$d"; cd "$d"
    hipcc $FLAGS -save-temps -c -o "$OBJ/$o" "$SRC/$src" -I"$SRC" "$@" $EXTRA 2>/dev/null
    cp "$d"/*-hip-amdgcn-amd-amdhsa-gfx950.s "$ASM/${o%.o}.s"
  else
    cd "$SRC"; hipcc $FLAGS -c -o "$OBJ/$o" "$src" "$@" $EXTRA
  fi' < "$OBJ/jobs"
hipcc ${LINKFLAGS} -shared -o "$OUT" "$OBJ"/*.o
if [ -n "$ASM" ]; then
  # which sources and which code generation the assembly belongs to (bench.kernel_source_hash(): the same bytes)
  H=$(for f in $(LC_ALL=C ls | grep -E '^(afsk_demod.*\.(h|hip)|afsk_kernels\.h|build\.sh)$' | LC_ALL=C sort); do printf %s "$f"; cat "$f"; done | sha256sum | cut -c1-16)
  echo "$H $CODEGEN" > "$ASM/BUILD_INFO"
fi
echo "built $(pwd)/$OUT ($CODEGEN code generation)"
