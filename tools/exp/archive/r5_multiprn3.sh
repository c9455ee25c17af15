#!/bin/bash
# round 5: the multi-PRN launch pairs as the default (buffer budget = 60 % of the free device memory) against one PRN per pair
# (BDS_ACQ_NOMULTI=1, hooks build), whole cfg3 calls alternating; then the acquisition parity tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"
run() {
  env "$@" timeout 300 python bench.py --workload b1c --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a ${COLD:---no-cold} 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; r=j['roofline']; print(tag.ljust(28), 'ms/step', round(j['ms_per_step'],2), 'search', round(s['search_ms'],2), 'frac', round(r['frac'],4), 'pair_ms', round(r['pair_ms'],3), r['kernel'][-40:], 'traffic', r['traffic'], 'sha', str(j['config'].get('results_sha256'))[8:20], (j.get('cold') or {}).get('b1c'))
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*"
}
{ for rep in 1 2; do run A=1; run BDS_ACQ_NOMULTI=1; done; COLD=" " run A=1; COLD=" " run BDS_ACQ_NOMULTI=1; } 2>&1 | tee gpurun_out/r05_multiprn_default_ab.txt
unset BDS_LIB_PATH
if [ "${TESTS:-1}" = 1 ]; then
timeout 2400 python -m pytest tests/test_acq_gpu.py tests/test_fullsize_gpu.py tests/test_sieve_gpu.py tests/test_sieve_stress_gpu.py tests/test_chain_gpu.py tests/test_multi_gpu.py tests/test_bench_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -8
fi
