#!/bin/bash
# round 5: cfg3 with several PRNs per launch pair (the cell-list mode of the small grids, BDS_ACQ_MULTI_ANY=1): the row workgroups
# of different PRNs walk the same spectrum rows at about the same time -- does the signal spectrum then come out of L2 / MALL
# instead of HBM (2.5 of the pair's 12.8 GB), and does that show in the time per cell?  Hooks build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"
P=${PRNS:-8}
run() {
  env "$@" timeout 300 python bench.py --workload b1c --prns $P --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a --no-cold 2>&1 | python -c "
import sys,json
tag=sys.argv[1]; P=int(sys.argv[2])
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; r=j['roofline']; print(tag.ljust(70), 'search', round(s['search_ms'],2), 'us/cell', round(s['search_ms']*1e3/(P*201),2), 'pair', round(r.get('pair_ms') or 0,3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), 'det', len(j['config']['satellites_detected']), 'sha', str(j['config'].get('results_sha256'))[8:20])
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*" $P
}
{
run A=1
run BDS_ACQ_MULTI_ANY=1 BDS_ACQ_PBCAP_GB=64 BDS_ACQ_PBCELLS=402
run BDS_ACQ_MULTI_ANY=1 BDS_ACQ_PBCAP_GB=64 BDS_ACQ_PBCELLS=804
run BDS_ACQ_MULTI_ANY=1 BDS_ACQ_PBCAP_GB=64 BDS_ACQ_PBCELLS=1608
run A=1
run BDS_ACQ_MULTI_ANY=1 BDS_ACQ_PBCAP_GB=64 BDS_ACQ_PBCELLS=804
} 2>&1 | tee gpurun_out/r05_multiprn.txt
