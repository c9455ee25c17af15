#!/bin/bash
# The small GPU parity cases on the DEBUG build of the library (device-side bounds asserts on every code-table / IF-window /
# candidate-list index, csrc/bds_debug.h).  On the GPU box:
#   BDS_DEBUG=1 ./build.sh && tools/run_debug.sh
# tests/conftest.py fails the session when any check fired.
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
export BDS_LIB_PATH="$ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_debug.so"
cd "$ROOT" && python -m pytest tests/test_acq_gpu.py tests/test_sieve_gpu.py tests/test_golden_gpu.py tests/test_track_gpu.py tests/test_chain_gpu.py \
    tests/test_framesync.py tests/test_unpack.py -m gpu -x -q -p no:cacheprovider "$@"
