#!/bin/bash
# same-box A/B of the column passes: wave-private (default) vs round-2 tile kernel (BDS_ACQ_WCOLS=0), B1C (8 PRNs) and B2a
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { # tag, env..., -- args
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-fast-path $ARGS 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(40), 'ms/step', round(j['ms_per_step'],3), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), 'n_extra', r['n_extra'])
    elif 'amdgpu.ids' not in l and ('Error' in l or 'error' in l): print(l.rstrip())
" "$tag"
}
ARGS="--workload b1c --prns 8 --steps 3 --warmup 1"
run "b1c wave" A=1
run "b1c tile (round 2)" BDS_ACQ_WCOLS=0
for v in $(ls tools/variants/libbds_*.so 2>/dev/null); do run "b1c $v" BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v; done
ARGS="--workload b1c --prns 8 --steps 3 --warmup 1"
run "b1c 1024x3072 wave" BDS_ACQ_FORCE_L1L2=1024x3072
run "b1c 1024x3072 tile" BDS_ACQ_FORCE_L1L2=1024x3072 BDS_ACQ_WCOLS=0
ARGS="--workload b2a --steps 10 --warmup 2"
run "b2a default (tile at 256)" A=1
run "b2a 512x1280 wave" BDS_ACQ_FORCE_L1L2=512x1280
run "b2a 512x1280 tile" BDS_ACQ_FORCE_L1L2=512x1280 BDS_ACQ_WCOLS=0
ARGS="--workload b1c --steps 5 --warmup 1"
run "b1c full wave" A=1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
