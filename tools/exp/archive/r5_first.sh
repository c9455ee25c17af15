#!/bin/bash
# round 5, first GPU trip: HBM stream probe, tracking numerics experiment, tracking tests on the unchanged default
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe/stream_probe.hip -o gpurun_out/stream_probe && timeout 300 gpurun_out/stream_probe > gpurun_out/r5_stream_probe.txt 2>&1
timeout 1500 python tools/exp/r5_trk_prec.py > gpurun_out/r5_trk_prec.txt 2> gpurun_out/r5_trk_prec.err
timeout 600 python -m pytest tests/test_track_gpu.py -x -q -m gpu > gpurun_out/r5_trk_tests.log 2>&1
tail -3 gpurun_out/r5_trk_tests.log
cat gpurun_out/r5_stream_probe.txt | head -50
