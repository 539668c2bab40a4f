#!/bin/bash
# Builds afskmodem_amd/csrc/libafsk_amd.so for gfx950 (cross-compiles without a GPU).  The two
# instantiations of the demod kernel dominate the build: they compile in parallel.
set -euo pipefail
cd "$(dirname "$0")"
ARCH=${AFSK_ARCH:-gfx950}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=${ARCH} -Wall -Wno-unused-function"
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
pids=()
for f in afsk_demod_small afsk_demod_big afsk_demod afsk_capi afsk_synth afsk_gate; do
  hipcc ${FLAGS} -c -o "$OBJ/$f.o" $f.hip "$@" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc ${FLAGS} -shared -o libafsk_amd.so "$OBJ"/*.o
echo "built $(pwd)/libafsk_amd.so"
