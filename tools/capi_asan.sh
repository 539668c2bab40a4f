#!/bin/bash
# CPU-only sanitizer run of the C-ABI's HOST code (afsk_capi.hip: argument checks, the I/O thread pool and its
# fork handler, the RIFF chunk walk, afsk_file_sizes): afsk_capi.hip is compiled with
# -fsanitize=address,undefined for the host side only (-fno-gpu-sanitize; GPU ASan is not available on the
# pool), the kernel launchers are stubbed (they need a device anyway), and the host-only tests run against
# the result through AFSK_AMD_LIB.  The ingest / upload / host-buffer entries need a GPU and are not covered.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=$(mktemp -d); trap 'rm -rf "$W"' EXIT
cat > "$W/stubs.hip" <<S
#include "$ROOT/afskmodem_amd/csrc/afsk_kernels.h"
namespace afsk {
hipError_t launch_gate(const GateArgs&, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_demod(const DemodArgs&, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_demod_uniform(const DemodArgs&, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_modulate(ModulateArgs, int32_t, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_noise(NoiseArgs, int32_t, hipStream_t) { return hipErrorUnknown; }
}
S
F="-std=c++17 -fPIC --offload-arch=${AFSK_ARCH:-gfx950} -Wno-unused-function"
hipcc -O1 -g $F -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -c -o "$W/capi.o" "$ROOT/afskmodem_amd/csrc/afsk_capi.hip"
hipcc -O1 $F -c -o "$W/stubs.o" "$W/stubs.hip"
hipcc -fPIC --offload-arch=${AFSK_ARCH:-gfx950} -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -shared -o "$W/libafsk_amd_asan.so" "$W/capi.o" "$W/stubs.o"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT AFSK_AMD_LIB="$W/libafsk_amd_asan.so" \
  python -m pytest tests/test_wav_probe_fuzz.py tests/test_host_api.py -q -k "wav or riff or file_sizes or fork or argument"
# ThreadSanitizer pass over the I/O pool: three Python threads call afsk_wav_probe concurrently (jobs serialise on
# the pool, workers are shared)
hipcc -O1 -g $F -fsanitize=thread -fno-gpu-sanitize -c -o "$W/capi_t.o" "$ROOT/afskmodem_amd/csrc/afsk_capi.hip" 2>/dev/null
hipcc -fPIC --offload-arch=${AFSK_ARCH:-gfx950} -fsanitize=thread -shared-libsan -shared -o "$W/libafsk_amd_tsan.so" "$W/capi_t.o" "$W/stubs.o"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" LD_PRELOAD=$RT AFSK_AMD_LIB="$W/libafsk_amd_tsan.so" python - <<'PY'
import sys, tempfile, threading, wave
sys.path.insert(0, ".")
from afskmodem_amd import batch
d = tempfile.mkdtemp()
names = []
for i in range(600):
    fn = f"{d}/a{i}.wav"
    with wave.open(fn, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(48000); f.writeframes(bytes(200 + i))
    names.append(fn)
def work(k):
    for _ in range(20):
        assert (batch.wav_probe(names[k::3])[2] == 0).all()
ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
print("tsan pass ok (any report would have ended the process with code 66)")
PY
# r4: the GPU-needing host entries (afsk_wav_ingest's staging ring, afsk_wav_upload, the host-buffer entries, the
# group plan) against the fake HIP runtime of tests/helpers/ -- asynchronous copies on worker threads, so a staging
# buffer reused before its copy ran is a race TSan reports -- under ThreadSanitizer and AddressSanitizer
for san in thread address,undefined; do
  tag=${san%%,*}
  bash "$ROOT/tests/helpers/build_stub_lib.sh" "$W/libafsk_stub_$tag.so" -fsanitize=$san -fno-omit-frame-pointer
  case $tag in thread) RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1);;
               *) RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1);; esac
  TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT \
    AFSK_STUB_LIB="$W/libafsk_stub_$tag.so" python -m pytest tests/test_capi_host_logic.py -q -p no:cacheprovider
done
