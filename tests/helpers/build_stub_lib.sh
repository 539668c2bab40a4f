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
#include <mutex>
namespace afsk {
hipError_t launch_gate(const GateArgs&, hipStream_t) { return hipSuccess; }
// the demod launchers remember what they were asked to launch ("device" memory is host memory here): the test reads
// the walk order of a plan back through afsk_stub_last_launch
static DemodArgs g_last; static int g_last_kind = 0;        // 1 = per-stream kernel, 2 = uniform kernel
static std::mutex g_last_mu;                                // (host entries are called from several threads at once)
hipError_t launch_demod(const DemodArgs& a, hipStream_t) { std::lock_guard<std::mutex> lk(g_last_mu); g_last = a; g_last_kind = 1; return hipSuccess; }
hipError_t launch_demod_uniform(const DemodArgs& a, hipStream_t) { std::lock_guard<std::mutex> lk(g_last_mu); g_last = a; g_last_kind = 2; return hipSuccess; }
hipError_t launch_modulate(ModulateArgs, int32_t, hipStream_t) { return hipSuccess; }
hipError_t launch_noise(NoiseArgs, int32_t, hipStream_t) { return hipSuccess; }
}
extern "C" int afsk_stub_last_launch(int32_t* out_index, int32_t cap, int32_t* out_uniform_bf, int32_t* out_n) {
    std::lock_guard<std::mutex> lk(afsk::g_last_mu);
    if (out_uniform_bf) *out_uniform_bf = afsk::g_last.uniform_bit_frames;
    if (out_n) *out_n = afsk::g_last.n_streams;
    if (!afsk::g_last.stream_index) return -afsk::g_last_kind;          // no index list: stream order
    for (int32_t k = 0; k < cap && k < afsk::g_last.n_streams; k++) out_index[k] = afsk::g_last.stream_index[k];
    return afsk::g_last_kind;
}
S
F="-O1 -g -std=c++17 -fPIC --offload-arch=${AFSK_ARCH:-gfx950} -Wno-unused-function -fno-gpu-sanitize"
hipcc $F "$@" -c -o "$W/capi.o" "$ROOT/afskmodem_amd/csrc/afsk_capi.hip"
hipcc $F -c -o "$W/stubs.o" "$W/stubs.hip"
hipcc $F "$@" -x hip -c -o "$W/rt.o" "$HERE/hip_stub_runtime.cpp"
SAN=""; for a in "$@"; do case $a in -fsanitize=*) SAN="$a -shared-libsan";; esac; done
hipcc -fPIC --offload-arch=${AFSK_ARCH:-gfx950} -fno-gpu-sanitize $SAN -shared -Wl,-Bsymbolic -o "$OUT" "$W/capi.o" "$W/stubs.o" "$W/rt.o"
