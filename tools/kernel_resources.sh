#!/bin/bash
# Per-kernel register / LDS / spill report of one translation unit (no GPU needed):
#   tools/kernel_resources.sh bds_acq.hip [grep pattern] [extra hipcc flags]
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/bds-3-b1c-b2a-sdr-receiver_amd/csrc"
f="$1"; pat="${2:-.}"; shift; shift
contract=off
[ "$f" = bds_acq.hip ] && contract="fast -fno-slp-vectorize"
[ "$f" = bds_track.hip ] && contract="off -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$SRC" -ffp-contract=$contract \
    --cuda-device-only -Rpass-analysis=kernel-resource-usage -c "$SRC/$f" -o /dev/null "$@" 2>&1 |
  sed -n 's/.*remark: *\(.*\) \[-Rpass-analysis=kernel-resource-usage\]$/\1/p' |
  awk -F': ' '/^Function Name/{name=$2} /^TotalSGPRs/{s=$2} /^VGPRs:/{v=$2} /^AGPRs/{a=$2} /^ScratchSize/{sc=$2} /^Occupancy/{o=$2} /^VGPRs Spill/{vs=$2}
              /^LDS Size/{print name, "| VGPR", v, "AGPR", a, "SGPR", s, "scratch", sc, "vspill", vs, "occ", o, "LDS", $2}' |
  while read -r name rest; do echo "$(echo "$name" | c++filt | sed 's/(.*//' | cut -c1-100) $rest"; done | grep -E "$pat"
