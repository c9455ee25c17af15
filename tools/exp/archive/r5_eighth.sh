#!/bin/bash
# round 5, eighth GPU trip: the merged-component f64 sums (k_corr<NC, FM, KIND>): unit test + acquisition parity, cfg2 / cfg3 stage
# times and the cfg2 timeline; then the LDS-DMA A/B of the column pass and the sustained power runs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_corr_gpu.py tests/test_acq_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py tests/test_chain_gpu.py tests/test_sieve_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r5_tests8.log 2>&1
tail -12 gpurun_out/r5_tests8.log
for w in b2a b1c; do
  rm -rf gpurun_out/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$w -o $w -- python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-cold --no-b2a --no-tracking-full > gpurun_out/prof_$w.log 2>&1
  db=$(find gpurun_out/prof_$w -name "*_results.db" | head -1)
  python tools/rocprof_summary.py "$db" > gpurun_out/r05_${w}_kernel_stats_corr.txt
  python tools/rocprof_timeline.py "$db" 34 > gpurun_out/r05_${w}_timeline_corr.txt
  find gpurun_out/prof_$w -name "*.db" -size +20M -delete
  grep "k_corr\|k_ref" gpurun_out/r05_${w}_kernel_stats_corr.txt | cut -c1-150
  grep '^{' gpurun_out/prof_$w.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$w under rocprof: ms/step', round(j['ms_per_step'],3), j['stage_ms'])"
done
cat gpurun_out/r05_b2a_timeline_corr.txt | cut -c1-150
for w in b2a b1c; do
timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-tracking --no-strict-f32 --no-cold --no-b2a --no-tracking-full 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$w plain: ms/step', round(j['ms_per_step'],3), j['stage_ms'])"
done
if [ "${AB:-1}" = 1 ]; then
VARIANTS="coldma" OUT=r05_coldma_ab.txt PRNS=8 bash tools/exp/r5_ab.sh
bash tools/exp/r5_power.sh 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['label'].ljust(70), 'ms/call', round(j['ms_per_call'],1), 'pair', round(j['pair_ms_mean'],3), 'P', j['power_W'], 'sclk', j['sclk_MHz'], 'n', j['samples_under_load'], j['source'][-80:])
    else: print(l.rstrip()[:150])
"
fi
