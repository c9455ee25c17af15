#!/bin/bash
# round 5, third GPU trip: whole GPU suite on the hooks build + release-build test, default bench line (cold leg, both tracking modes)
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r5_gpu_tests.log 2>&1
tail -25 gpurun_out/r5_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r5_bench_b1c.json 2> gpurun_out/r5_bench_b1c.err
tail -c 3000 gpurun_out/r5_bench_b1c.json
