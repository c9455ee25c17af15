#!/bin/bash
# power / clock samples (rocm-smi) while the cfg3 search runs: is the chip at its power cap?
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | grep -v "^$" | head -30
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' ; echo; sleep 0.5; done ) > gpurun_out/r04_power_samples.txt &
SMI=$!
BDS_ACQ_CLOCKPROBE=1 timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --workload b1c --prns 63 --steps 3 --warmup 1 2>&1 | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']; print('ms/step', j['ms_per_step'], 'pair', r['pair_ms'], 'clock', r.get('valu',{}).get('shader_clock_GHz'))"
wait $SMI
cat gpurun_out/r04_power_samples.txt | sed 's/GPU\[0\]\s*: //g' | cut -c1-220
