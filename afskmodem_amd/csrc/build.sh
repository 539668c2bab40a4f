#!/bin/bash
# Builds afskmodem_amd/csrc/libafsk_amd.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
ARCH=${AFSK_ARCH:-gfx950}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=${ARCH} -Wall -Wno-unused-function"
hipcc ${FLAGS} -shared -o libafsk_amd.so afsk_capi.hip afsk_demod.hip afsk_synth.hip afsk_gate.hip "$@"
echo "built $(pwd)/libafsk_amd.so"
