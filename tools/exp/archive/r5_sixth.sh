#!/bin/bash
# round 5, sixth GPU trip: kernel trace of cfg2 (what the device refinement chain costs), strict-carrier variants of the
# tracking correlator (prec 4 / 5), sustained power with the clock-ceiling and lower-activity A/Bs, fp32-storage sieve error
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_b2a
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b2a -o b2a -- python bench.py --workload b2a --steps 20 --warmup 2 --no-cpu-baseline --no-tracking --no-strict-f32 --no-cold > gpurun_out/prof_b2a.log 2>&1
db=$(find gpurun_out/prof_b2a -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$db" > gpurun_out/r05_b2a_kernel_stats.txt
find gpurun_out/prof_b2a -name "*.db" -size +20M -delete
head -30 gpurun_out/r05_b2a_kernel_stats.txt | cut -c1-200
timeout 1500 python tools/exp/r5_trk_prec.py 4 5 > gpurun_out/r5_trk_prec5.txt 2> gpurun_out/r5_trk_prec5.err
python - <<'PY'
import json
for l in open('gpurun_out/r5_trk_prec5.txt'):
    r = json.loads(l)
    if 'fixture' in r:
        print(r['fixture'], 'prec', r['prec'], 'seg', r['seg'], round(r['us_per_epoch'], 1), [(c['first_bad_iq_1e-4'], c['first_bad_code_1e-6'], c['max_remCode']) for c in r['channels']])
    else:
        print(r)
PY
timeout 600 python tools/sieve_error.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_sieve_error_small.txt
cat gpurun_out/r05_sieve_error_small.txt
bash tools/exp/r5_power.sh 2>&1 | cut -c1-900
