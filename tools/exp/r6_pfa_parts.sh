#!/bin/bash
# Where does the N-point pair's time go?  The probe (tools/probe/pfa_pair.hip) built with parts of the kernels compiled out
# (csrc/bds_acq_pfa.h, PFA_EXP_*: results INVALID, timing only).  Build here (no GPU needed), run on the GPU box:
#   tools/exp/r6_pfa_parts.sh build ; gpurun -- tools/exp/r6_pfa_parts.sh run > profiles/r06_pfa53_parts.txt
cd "$(dirname "$0")/../.."
VARS="base C_NOEXACT C_NOMFMA C_NOEPI C_NOLOAD R_NOLOAD R_NOSTORE R_NOBAR"
if [ "$1" = build ]; then
    for v in $VARS; do
        d=""; [ $v != base ] && d="-DPFA_EXP_$v"
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize $d ${PFA_EXTRA:-} -Ibds-3-b1c-b2a-sdr-receiver_amd/csrc -Iinclude \
            tools/probe/pfa_pair.hip -o tools/probe/pfa_pair_$v.bin || exit 1
    done
else
    for v in $VARS; do echo "== $v"; tools/probe/pfa_pair_$v.bin ${PRNS:-8} 3 | grep timing | tail -1; done
fi
