#!/bin/bash
# Round-6 measurement pass (GPU box): default bench lines, kernel-trace stats (cfg3, cfg2, tracking), PMC counters of the N-point pair.
# Everything lands under gpurun_out/; tools/r6_collect.sh copies the summaries into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/bench_b1c.json 2> gpurun_out/bench_b1c.err; echo "bench b1c rc=$?"
timeout 600 python bench.py --workload b2a > gpurun_out/bench_b2a.json 2> gpurun_out/bench_b2a.err; echo "bench b2a rc=$?"
bash tools/profile_run.sh > gpurun_out/profile_run.log 2>&1; tail -3 gpurun_out/profile_run.log
bash tools/profile_track.sh > gpurun_out/profile_track.log 2>&1; tail -3 gpurun_out/profile_track.log
# (8 PRNs = one launch pair of the library default: 1608 cells; PMC_CELLS of tools/r6_collect.sh says the same)
BENCH_ARGS="--workload b1c --steps 1 --warmup 0 --no-cpu-baseline --no-tracking --no-strict-f32 --no-b2a --no-cold --prns 8" bash tools/pmc_run.sh > gpurun_out/pmc_run.log 2>&1; tail -2 gpurun_out/pmc_run.log
cp gpurun_out/pmc_summary.txt gpurun_out/pmc_summary_default.txt
