#!/bin/bash
# round 5: whole cfg3 calls (63 PRNs) with 8 / 16 / 21 PRNs per launch pair against the default (one PRN's 201 cells per pair)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"
run() {
  env "$@" timeout 300 python bench.py --workload b1c --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a --no-cold 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; r=j['roofline']; print(tag.ljust(70), 'ms/step', round(j['ms_per_step'],2), 'search', round(s['search_ms'],2), 'us/cell', round(s['search_ms']*1e3/(63*201),3), 'frac', round(r['frac'],4), 'det', len(j['config']['satellites_detected']), 'sha', str(j['config'].get('results_sha256'))[8:20])
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*"
}
{
for rep in 1 2; do
run A=1
for c in ${CELLS:-1608 3216 4221}; do run BDS_ACQ_MULTI_ANY=1 BDS_ACQ_PBCAP_GB=${CAP:-128} BDS_ACQ_PBCELLS=$c; done
done
} 2>&1 | tee gpurun_out/${OUT:-r05_multiprn_full.txt}
