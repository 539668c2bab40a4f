#!/bin/bash
# Build a variant of libafsk_amd.so from a patched copy of the kernel sources, for kbench A/B runs:
#   tools/build_variant.sh <name> [sed-expression applied to csrc/*.h csrc/*.hip ...]
# -> tools/libafsk_<name>.so   (KBENCH_LIB_B=tools/libafsk_<name>.so ./kbench ...)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
W=/tmp/afsk_variant_$name
rm -rf "$W"; mkdir -p "$W/afskmodem_amd" "$W/include"
cp -r "$ROOT/afskmodem_amd/csrc" "$W/afskmodem_amd/csrc"
cp "$ROOT/include/afsk_amd.h" "$W/include/"
rm -f "$W/afskmodem_amd/csrc/"*.so
for e in "$@"; do sed -i "$e" "$W"/afskmodem_amd/csrc/*.h "$W"/afskmodem_amd/csrc/*.hip; done
(cd "$W/afskmodem_amd/csrc" && bash build.sh >/dev/null)
cp "$W/afskmodem_amd/csrc/libafsk_amd.so" "$ROOT/tools/libafsk_$name.so"
echo "built tools/libafsk_$name.so"
# KBENCH=1: also build tools/kbench_<name> against the patched sources (its in-tool kernels -- timeline
# stamps, skip_sync / skip_valu ablations -- then come from the variant too)
if [ "${KBENCH:-0}" = "1" ]; then
  mkdir -p "$W/tools"
  cp "$ROOT/tools/kbench.hip" "$ROOT/tools/afsk_twopass.h" "$W/tools/"
  (cd "$W/tools" && hipcc -O3 -std=c++17 --offload-arch=${AFSK_ARCH:-gfx950} -Wno-unused-function ${KBENCH_DEFS:-} -o "$ROOT/tools/kbench_$name" kbench.hip \
      ../afskmodem_amd/csrc/afsk_synth.hip ../afskmodem_amd/csrc/afsk_gate.hip -ldl)
  echo "built tools/kbench_$name"
fi
