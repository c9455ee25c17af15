#!/bin/bash
# Round-4 measurement pass (GPU box): PMC counters, kernel-trace stats (cfg3, cfg2, tracking), phase-clock profile, co-issue
# probe, sieve stress, plan experiment (729 x 4096 against 768 x 4096 on the run-time-plan kernels), power samples, default
# bench line.  Everything lands under gpurun_out/; tools/r4_collect.sh copies the summaries into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/pmc_run.sh > gpurun_out/pmc_run.log 2>&1; tail -2 gpurun_out/pmc_run.log
cp gpurun_out/pmc_summary.txt gpurun_out/pmc_summary_default.txt
bash tools/profile_run.sh > gpurun_out/profile_run.log 2>&1; tail -3 gpurun_out/profile_run.log
bash tools/profile_track.sh > gpurun_out/profile_track.log 2>&1; tail -3 gpurun_out/profile_track.log
BDS_LIB_PATH=tools/variants/libbds_phases.so timeout 300 python tools/phases.py --prns 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_phases.txt; head -3 gpurun_out/r04_phases.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probe/coissue.hip -o /tmp/coissue && /tmp/coissue > gpurun_out/r04_coissue.txt 2>&1
timeout 900 python tools/sieve_stress.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_sieve_error.txt; tail -2 gpurun_out/r04_sieve_error.txt
# item 8: the shorter 3^6 x 2^12 transform against the 3 x 2^20 one, both on the run-time-plan kernels (no specialised 729-point pass exists)
for pl in 768x4096 729x4096; do
  BDS_ACQ_GENERIC=1 BDS_ACQ_FORCE_L1L2=$pl timeout 600 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a --workload b1c --prns 2 --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('plan $pl (run-time-plan kernels): pair', round(r['pair_ms'],3), 'ms per 201 cells, fft_len', j['config']['fft_len'], 'detected', j['config']['satellites_detected'])"
done > gpurun_out/r04_plan_729.txt 2>&1; cat gpurun_out/r04_plan_729.txt
bash tools/exp/r4_power.sh > gpurun_out/r04_power.txt 2>&1; grep -o "Power (W): [0-9.]*\|sclk clock level: [0-9S]: ([0-9]*Mhz)" gpurun_out/r04_power_samples.txt | paste - - | sort | uniq -c | sort -rn | head -5
timeout 1200 python bench.py > gpurun_out/bench_b1c.json 2> gpurun_out/bench_b1c.err; echo "bench b1c rc=$?"
timeout 600 python bench.py --workload b2a > gpurun_out/bench_b2a.json 2> gpurun_out/bench_b2a.err; echo "bench b2a rc=$?"
