#!/bin/bash
# round 5: a few PRNs per launch pair (2 / 4 / 8: 10 / 20 / 40 GB of inter-pass buffer) with 67-bin row chunks, so that a launch has as many
# row workgroups as the lean mode's -- does sharing the spectrum rows between 2-8 PRNs pay without the 150 GiB buffer?  Hooks build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"
run() {
  env "$@" timeout 300 python bench.py --lean --workload b1c --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a --no-cold 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; r=j['roofline']; print(tag.ljust(44), 'ms/step', round(j['ms_per_step'],2), 'frac', round(r['frac'],4), 'pair', round(r['pair_ms'],3), r['kernel'][-32:], 'sha', str(j['config'].get('results_sha256'))[8:20])
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*"
}
{ for rep in 1 2; do run A=1; for gb in 11 21 42; do run BDS_ACQ_PBCAP_GB=$gb BDS_ACQ_LIST_GC=67; done; run BDS_ACQ_PBCAP_GB=21; done; } 2>&1 | tee gpurun_out/r05_small_pairs.txt
