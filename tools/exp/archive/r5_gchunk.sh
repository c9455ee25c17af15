#!/bin/bash
# round 5: cells per row workgroup in the lean mode (one PRN per launch pair): 768 rows x nch chunks must be a whole number of
# 512-workgroup waves (nch even); 34 cells (6 chunks) was tuned in round 3 -- still the best after rounds 4-5?  Hooks build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"
run() {
  env "$@" timeout 300 python bench.py --lean --workload b1c --prns 16 --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a --no-cold 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; r=j['roofline']; print(tag.ljust(28), 'search', round(s['search_ms'],2), 'pair', round(r['pair_ms'],4), 'rows', round(r['rows_ms'],4), 'cols', round(r['cols_ms'],4), 'sha', str(j['config'].get('results_sha256'))[8:20])
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*"
}
{ for rep in 1 2; do for g in 34 26 51 101 21; do run BDS_ACQ_GCHUNK=$g; done; done; } 2>&1 | tee gpurun_out/r05_gchunk.txt
