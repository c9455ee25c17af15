#!/bin/bash
# bench line per search arithmetic / storage mode (few PRNs): default (fp32 arithmetic on fp16 storage),
# fp32 storage, packed-fp16 arithmetic.  Extra env assignments can be passed as arguments (applied to every mode).
cd "$GRAFT_REPO_ROOT"
P=${PRNS:-6}
W=${WORKLOAD:-b1c}
for m in "BDS_X=0" "BDS_ACQ_FP16=0" "BDS_ACQ_HMATH=1"; do
  echo "== mode: $m $*"
  env $m "$@" timeout 600 python bench.py --workload $W --prns $P --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-fast-path 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['dtype'], j['roofline'].get('storage'), 'ms/step', round(j['ms_per_step'],2), 'stage', {k:round(v,2) for k,v in j['stage_ms'].items()}, 'pair_ms', round(r['pair_ms'],3), 'rows/cols', r.get('rows_ms'), r.get('cols_ms'), 'frac', round(r['frac'],3), 'extra', r.get('n_extra'), 'det', j['config']['satellites_detected'])
    elif 'amdgpu.ids' not in l: print(l.rstrip())
"
done
