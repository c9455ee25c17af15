#!/bin/bash
# What the column pass's waves wait for: SQ counters of the probe's variants (tools/exp/r6_pfa_parts.sh build first) under rocprofv3 --pmc (GPU box)
#   tools/exp/r6_pfa_parts_pmc.sh > profiles/r06_pfa53_parts_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${VARS:-base C_NOEXACT C_NOLOAD C_NOEPI}; do
    rm -rf gpurun_out/pmcp
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
        -d gpurun_out/pmcp -o p -- tools/probe/pfa_pair_$v.bin 8 1 > gpurun_out/pmcp_$v.log 2>&1
    echo "#### $v (rc=$?)"
    python tools/pmc_summary.py gpurun_out/pmcp/p_results.db 2>/dev/null | awk '/k_pfa_cols<2, false>/,/~duration/'
done
rm -rf gpurun_out/pmcp
