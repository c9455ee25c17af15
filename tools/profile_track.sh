#!/bin/bash
# kernel-trace profile of the tracking loop (tools/bench_track.py), per mode
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in WB B2A; do
  rm -rf gpurun_out/prof_trk_$m
  ep=100; [ $m = B2A ] && ep=1000
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_trk_$m -o t -- python tools/bench_track.py --mode $m --epochs $ep > gpurun_out/prof_trk_$m.log 2>&1
  tail -1 gpurun_out/prof_trk_$m.log | cut -c1-300
  python tools/rocprof_summary.py $(find gpurun_out/prof_trk_$m -name "*_results.db" | head -1) > gpurun_out/kernel_stats_trk_$m.txt
  head -6 gpurun_out/kernel_stats_trk_$m.txt
  find gpurun_out/prof_trk_$m -name "*.db" -size +20M -delete
done
