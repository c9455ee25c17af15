#!/bin/bash
# Kernel-trace profiles of the default bench commands (run on the GPU box through gpurun):
#   bash tools/profile_run.sh            -> gpurun_out/prof_b1c, gpurun_out/prof_b2a + text summaries
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in b1c b2a; do
  rm -rf gpurun_out/prof_$w
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$w -o $w -- python bench.py --workload $w --no-cpu-baseline --no-tracking --no-strict-f32 --no-b2a > gpurun_out/prof_$w.log 2>&1
  echo "$w rc=$?"
  db=$(find gpurun_out/prof_$w -name "*_results.db" | head -1)
  python tools/rocprof_summary.py "$db" > gpurun_out/kernel_stats_$w.txt
  grep -E "^\{" gpurun_out/prof_$w.log > gpurun_out/bench_under_rocprof_$w.json
  find gpurun_out/prof_$w -name "*.db" -size +20M -delete
  head -8 gpurun_out/kernel_stats_$w.txt
done
