#!/bin/bash
# round 5, second GPU trip: the strict carrier with hand-written division / sincos (prec 4) against the library form (prec 3)
# and the fp32 form (prec 0): first flip over the full horizons, us per epoch; tracking tests on the new default
mkdir -p gpurun_out
timeout 1200 python tools/exp/r5_trk_prec.py 0 3 4 > gpurun_out/r5_trk_prec4.txt 2> gpurun_out/r5_trk_prec4.err
grep timing gpurun_out/r5_trk_prec4.txt
timeout 900 python -m pytest tests/test_track_gpu.py tests/test_track_long_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/r5_trk_tests.log 2>&1
tail -5 gpurun_out/r5_trk_tests.log
