#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
df -h /tmp | tail -1; free -g | head -2 | tail -1; nproc
timeout 1500 python -m pytest tests/test_multi_gpu.py tests/test_cfg4_gpu.py tests/test_bench_gpu.py -x -q > gpurun_out/tests2.log 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/tests2.log | tail -15
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 3000 gpurun_out/bench_default.json
