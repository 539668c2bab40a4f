#!/bin/bash
# Test-only build of afsk_capi.hip's HOST code against the fake HIP runtime (hip_stub_runtime.cpp) and stub kernel
# launchers:   build_stub_lib.sh <out.so> [extra hipcc flags, e.g. -fsanitize=thread]
# The demod launchers "succeed" without writing anything (outputs stay as the caller zeroed them), so the host
# entries run start to finish; nothing here is part of the product library.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
OUT=$1; shift
W=$(mktemp -d); trap 'rm -rf "$W"' EXIT
cat > "$W/stubs.hip" <<S
#include "$ROOT/afskmodem_amd/csrc/afsk_kernels.h"
namespace afsk {
hipError_t launch_gate(const GateArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_demod(const DemodArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_demod_uniform(const DemodArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_modulate(ModulateArgs, int32_t, hipStream_t) { return hipSuccess; }
hipError_t launch_noise(NoiseArgs, int32_t, hipStream_t) { return hipSuccess; }
}
S
F="-O1 -g -std=c++17 -fPIC --offload-arch=${AFSK_ARCH:-gfx950} -Wno-unused-function -fno-gpu-sanitize"
hipcc $F "$@" -c -o "$W/capi.o" "$ROOT/afskmodem_amd/csrc/afsk_capi.hip"
hipcc $F -c -o "$W/stubs.o" "$W/stubs.hip"
hipcc $F "$@" -x hip -c -o "$W/rt.o" "$HERE/hip_stub_runtime.cpp"
SAN=""; for a in "$@"; do case $a in -fsanitize=*) SAN="$a -shared-libsan";; esac; done
hipcc -fPIC --offload-arch=${AFSK_ARCH:-gfx950} -fno-gpu-sanitize $SAN -shared -Wl,-Bsymbolic -o "$OUT" "$W/capi.o" "$W/stubs.o" "$W/rt.o"
