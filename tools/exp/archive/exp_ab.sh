#!/bin/bash
# same-box A/B of the search kernels on cfg3 (8 PRNs): wave-private row / column passes (defaults) vs the round-2 kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 $ARGS 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(44), 'ms/step', round(j['ms_per_step'],3), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), 'n_extra', r['n_extra'], j['config']['satellites_detected'])
    elif 'amdgpu.ids' not in l and ('Error' in l or 'error' in l): print(l.rstrip())
" "$tag"; }
ARGS="--workload b1c --prns 8 --steps 3 --warmup 1"
run "wave rows + wave cols (default)" A=1
run "round-2 rows + wave cols" BDS_ACQ_WROWS=0
run "wave rows + tile cols" BDS_ACQ_WCOLS=0
run "round-2 rows + tile cols (round 2)" BDS_ACQ_WROWS=0 BDS_ACQ_WCOLS=0
for v in $(ls tools/variants/libbds_*.so 2>/dev/null); do run "$v" BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v; run "$v round-2 rows" BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v BDS_ACQ_WROWS=0; done
