#!/bin/bash
# phase-clock profile of the wave-private search kernels (tools/phases.py) + the valid co-issue probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out


BDS_LIB_PATH=tools/variants/libbds_phases.so timeout 300 python tools/phases.py --prns 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_phases.txt
