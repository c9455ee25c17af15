#!/bin/bash
# Build a timing variant of the library: tools/build_variant.sh NAME "-DFLAG ..." ["base flags override"]
#   -> tools/variants/libbds_NAME.so   (only bds_acq.hip is recompiled; the other objects come from the in-tree build)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
PKG="$ROOT/bds-3-b1c-b2a-sdr-receiver_amd"
mkdir -p "$ROOT/tools/variants"
O="$ROOT/tools/variants/bds_acq_$1.o"
BASE="${3:--ffp-contract=fast -fno-slp-vectorize}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -I"$ROOT/include" -I"$PKG/csrc" \
    $BASE $2 -c "$PKG/csrc/bds_acq.hip" -o "$O"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$PKG/build/bds_codes.o" "$PKG/build/bds_api.o" "$O" "$PKG/build/bds_track.o" "$PKG/build/bds_sync.o" \
    "$PKG/build/bds_multi.o" -ldl -o "$ROOT/tools/variants/libbds_$1.so" -Wl,-rpath,/opt/rocm/lib
rm -f "$O"
echo "built variant $1"
