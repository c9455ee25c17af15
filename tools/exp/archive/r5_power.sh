#!/bin/bash
# round 5: sustained power / clock under the cfg3 search (25 s loops, ~20 Hz samples) and the clock-ceiling A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
rocm-smi --showmaxpower --showperflevel --showsclkrange 2>&1 | grep -v "^$" | grep -v "====" | head -12
timeout 200 python tools/power_sustained.py --seconds 25 --label "default"
if [ "${CEIL:-0}" = 1 ]; then  # (on the pool's boxes rocm-smi accepts the ceiling -- rc 0 -- and nothing changes: 200.9 / 201.3 / 201.1 ms per call at 2100 / 1800 / 1500 MHz)
for mhz in 2100 1800 1500; do
  timeout 200 python tools/power_sustained.py --seconds 15 --label "sclk ceiling $mhz MHz" --sclk-max $mhz
done
rocm-smi --resetperfdeterminism 2>&1 | tail -2
rocm-smi --resetclocks 2>&1 | tail -2
fi
# lower-activity variants of the same search (more instructions for the same flops / more bytes): does the clock go up and the time stay?
timeout 200 python tools/power_sustained.py --seconds 15 --label "plain-fp32 butterflies (BDS_ACQ_PK=0)" --env BDS_ACQ_PK=0
timeout 200 python tools/power_sustained.py --seconds 15 --label "round-2 row + tile column kernels (BDS_ACQ_WROWS=0 BDS_ACQ_WCOLS=0)" --env BDS_ACQ_WROWS=0 --env BDS_ACQ_WCOLS=0
timeout 200 python tools/power_sustained.py --seconds 15 --label "fp32 storage (BDS_ACQ_FP16=0)" --env BDS_ACQ_FP16=0
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_power_sustained.txt
