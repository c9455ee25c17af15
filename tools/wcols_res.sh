#!/bin/bash
# register / scratch / occupancy of k_cols_wave_f<WC_S, 2, false, __half2> (no GPU needed); -S dumps the ISA to /tmp/wcols.s
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/bds-3-b1c-b2a-sdr-receiver_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$ROOT/include" -I"$SRC" -ffp-contract=fast -fno-slp-vectorize \
    --cuda-device-only -Rpass-analysis=kernel-resource-usage -S "$ROOT/tools/probe/wcols_tu.hip" -o /tmp/wcols.s "$@" 2>&1 |
    grep -E "VGPRs:|Scratch|Occupancy|VGPRs Spill|SGPRs:" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' '; echo
grep -c "v_\(fma\|add\|sub\|mul\|fmac\)_f32" /tmp/wcols.s | sed 's/^/f32 arith instrs: /'
