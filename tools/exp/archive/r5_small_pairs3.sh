#!/bin/bash
# round 5: the automatic choice of the row chunk (no BDS_ACQ_LIST_GC) at 4 / 8 / 16 / 32 PRNs per launch pair against the lean mode. Hooks build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"
run() {
  env "$@" timeout 300 python bench.py --lean --workload b1c --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a --no-cold 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; r=j['roofline']; print(tag.ljust(30), 'ms/step', round(j['ms_per_step'],2), 'frac', round(r['frac'],4), 'pair', round(r['pair_ms'],3), r['kernel'][-34:], 'sha', str(j['config'].get('results_sha256'))[8:20])
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*"
}
{ for rep in 1 2; do run A=1; for gb in 21 42 84 170; do run BDS_ACQ_PBCAP_GB=$gb; done; done; } 2>&1 | tee gpurun_out/r05_small_pairs3.txt
