#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/test_multi_gpu.py tests/test_cfg4_gpu.py tests/test_bench_gpu.py -x -q > gpurun_out/tests2.log 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/tests2.log | tail -15
