#!/bin/bash
# Build a timing variant of the library: tools/build_variant.sh NAME "-DFLAG ..." ["base flags override"]
#   -> tools/variants/libbds_NAME.so   (only bds_acq.hip -- or VFILE=bds_track: bds_track.hip -- is recompiled; the other objects come from the in-tree build)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
PKG="$ROOT/bds-3-b1c-b2a-sdr-receiver_amd"
mkdir -p "$ROOT/tools/variants"
F="${VFILE:-bds_acq}"   # VFILE=bds_track: rebuild the tracking object instead (contraction off, as build.sh)
O="$ROOT/tools/variants/${F}_$1.o"
BASE="${3:--ffp-contract=fast -fno-slp-vectorize}"
[ "$F" = bds_track ] && BASE="-ffp-contract=off -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -I"$ROOT/include" -I"$PKG/csrc" \
    $BASE $2 -c "$PKG/csrc/$F.hip" -o "$O"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$PKG/build/bds_codes.o" "$PKG/build/bds_api.o" $([ "$F" = bds_acq ] && echo "$O" || echo "$PKG/build/bds_acq.o") $([ "$F" = bds_track ] && echo "$O" || echo "$PKG/build/bds_track.o") "$PKG/build/bds_sync.o" \
    "$PKG/build/bds_multi.o" -ldl -o "$ROOT/tools/variants/libbds_$1.so" -Wl,-rpath,/opt/rocm/lib
rm -f "$O"
echo "built variant $1"
