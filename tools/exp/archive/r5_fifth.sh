#!/bin/bash
# round 5, fifth GPU trip: remaining parity tests on the device refinement chain, cfg2 / cfg3 stage times, rocFFT comparator, sustained power
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_sieve_gpu.py tests/test_golden_gpu.py tests/test_chain_gpu.py tests/test_fullsize_gpu.py tests/test_multi_gpu.py tests/test_sieve_stress_gpu.py tests/test_fuzz_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r5_refine_tests.log 2>&1
tail -15 gpurun_out/r5_refine_tests.log
for wl in b2a b1c; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-tracking --no-strict-f32 --no-cold 2> gpurun_out/r5_bench_$wl.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$wl', 'ms/step', round(j['ms_per_step'],3), j['stage_ms'], 'pair', j['roofline']['pair_ms'], 'sha', j['config'].get('results_sha256'))
"
done
hipcc --offload-arch=gfx950 -O3 tools/probe/rocfft_pair.hip -lrocfft -o gpurun_out/rocfft_pair && (cd /tmp && timeout 900 $GRAFT_REPO_ROOT/gpurun_out/rocfft_pair) > gpurun_out/r05_rocfft_baseline.txt 2>&1
cat gpurun_out/r05_rocfft_baseline.txt
bash tools/exp/r5_power.sh
