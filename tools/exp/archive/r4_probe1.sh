#!/bin/bash
# co-issue probe (tools/probe/coissue.hip)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probe/coissue.hip -o /tmp/coissue && /tmp/coissue > gpurun_out/r04_coissue.txt 2>&1
grep -v roles gpurun_out/r04_coissue.txt
