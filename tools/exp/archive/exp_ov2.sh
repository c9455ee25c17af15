#!/bin/bash
# column pass beside the next group's row pass (second stream) with 4-column tiles, whose workgroups fit next to two row-pass workgroups
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
source_run() { :; }
cd "$GRAFT_REPO_ROOT"
P=${PRNS:-8}
run() {
  env "$@" timeout 600 python bench.py --workload b1c --prns $P --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(tag.ljust(50), 'step', round(j['ms_per_step'],2), 'search', round(j['stage_ms']['search_ms'],2), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), 'det', j['config']['satellites_detected'])
    elif 'amdgpu.ids' not in l: print(l.rstrip())
" "$*"
}
run BDS_X=0
run BDS_ACQ_LOGT=2
run BDS_ACQ_OVERLAP=1
run BDS_ACQ_OVERLAP=1 BDS_ACQ_LOGT=2
run BDS_X=0
