#!/bin/bash
# round 5: cfg2 with the second-peak pass reading the main search's inter-pass buffer (default) against its own row pass
# (BDS_ACQ_NO_BWREUSE=1, hooks build), alternating on one box; then the cfg2 parity tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"
run() {
  env "$@" timeout 300 python bench.py --workload b2a --steps 20 --warmup 3 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-cold 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; print(sys.argv[1].ljust(28), 'ms/step', round(j['ms_per_step'],3), 'fwd', round(s['forward_ms'],3), 'search', round(s['search_ms'],3), 'refine', round(s['refine_ms'],3), 'sha', str(j['config'].get('results_sha256'))[8:24])
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*"
}
{ for rep in 1 2 3; do run A=1; run BDS_ACQ_NO_BWREUSE=1; done; } 2>&1 | tee gpurun_out/r05_b2a_reuse_ab.txt
unset BDS_LIB_PATH
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_acq_gpu.py tests/test_chain_gpu.py -m gpu -x -q -k "b2a or cfg2 or B2A or chain" 2>&1 | tail -4
