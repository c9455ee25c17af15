#!/bin/bash
# Round-3 measurement pass (GPU box): PMC counters (default and fp32 storage), kernel-trace stats, tracking profile, sieve error
# at the cfg3 plan, the default bench lines.  Everything lands under gpurun_out/; the summaries are copied to profiles/ by hand.
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_run.sh > gpurun_out/pmc_run.log 2>&1; tail -2 gpurun_out/pmc_run.log
cp gpurun_out/pmc_summary.txt gpurun_out/pmc_summary_default.txt
# fp32 storage: HBM traffic only
ARGS="--workload b1c --steps 1 --warmup 0 --no-cpu-baseline --no-tracking --no-strict-f32 --prns 2"
i=0
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT TCC_MISS"; do
  i=$((i+1))
  BDS_ACQ_FP16=0 timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmcf -o pass$i -- python bench.py $ARGS > gpurun_out/pmcf_pass$i.log 2>&1; echo "f32 pass$i rc=$?"
done
python tools/pmc_summary.py gpurun_out/pmcf/pass*_results.db > gpurun_out/pmc_summary_f32.txt 2>&1; rm -rf gpurun_out/pmcf
bash tools/profile_run.sh > gpurun_out/profile_run.log 2>&1; tail -3 gpurun_out/profile_run.log
bash tools/profile_track.sh > gpurun_out/profile_track.log 2>&1; tail -3 gpurun_out/profile_track.log
timeout 600 python tools/sieve_error_cfg3.py > gpurun_out/sieve_error_cfg3.txt 2> gpurun_out/sieve_error_cfg3.err; head -5 gpurun_out/sieve_error_cfg3.txt
timeout 900 python bench.py > gpurun_out/bench_b1c.json 2> gpurun_out/bench_b1c.err; echo "bench b1c rc=$?"
timeout 600 python bench.py --workload b2a > gpurun_out/bench_b2a.json 2> gpurun_out/bench_b2a.err; echo "bench b2a rc=$?"
