#!/bin/bash
# round 5, fourth GPU trip: the device refinement chain (parity tests, cfg2 / cfg3 stage times), then the cache-policy A/B of the cfg3 pair
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_acq_gpu.py tests/test_sieve_gpu.py tests/test_golden_gpu.py tests/test_chain_gpu.py tests/test_fullsize_gpu.py tests/test_multi_gpu.py tests/test_sieve_stress_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/r5_refine_tests.log 2>&1
tail -15 gpurun_out/r5_refine_tests.log
for wl in b2a b1c; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-tracking --no-strict-f32 --no-cold 2> gpurun_out/r5_bench_$wl.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$wl', 'ms/step', round(j['ms_per_step'],3), j['stage_ms'], 'sha', j['config'].get('results_sha256'))
"
done
bash tools/exp/r5_nt.sh
