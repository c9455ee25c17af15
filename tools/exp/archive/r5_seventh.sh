#!/bin/bash
# round 5, seventh GPU trip: parity on the reworked chain + strict tracking default (prec 5), kernel traces of cfg2 and cfg3, LDS-DMA A/B, power with variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_acq_gpu.py tests/test_fullsize_gpu.py tests/test_track_gpu.py tests/test_track_long_gpu.py tests/test_cfg4_gpu.py tests/test_golden_gpu.py tests/test_chain_gpu.py -x -q -m gpu > gpurun_out/r5_tests7.log 2>&1
tail -8 gpurun_out/r5_tests7.log
for w in b2a b1c; do
  rm -rf gpurun_out/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$w -o $w -- python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-tracking --no-strict-f32 --no-cold --no-b2a > gpurun_out/prof_$w.log 2>&1
  db=$(find gpurun_out/prof_$w -name "*_results.db" | head -1)
  python tools/rocprof_summary.py "$db" > gpurun_out/r05_${w}_kernel_stats.txt
  find gpurun_out/prof_$w -name "*.db" -size +20M -delete
  head -32 gpurun_out/r05_${w}_kernel_stats.txt | cut -c1-170
  grep '^{' gpurun_out/prof_$w.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$w under rocprof: ms/step', round(j['ms_per_step'],3), j['stage_ms'])"
done
VARIANTS="coldma" OUT=r05_coldma_ab.txt PRNS=8 bash tools/exp/r5_ab.sh
bash tools/exp/r5_power.sh 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['label'].ljust(70), 'ms/call', round(j['ms_per_call'],1), 'pair', round(j['pair_ms_mean'],3), 'P', j['power_W'], 'sclk', j['sclk_MHz'], 'n', j['samples_under_load'], j['source'][-80:])
    else: print(l.rstrip()[:150])
"
