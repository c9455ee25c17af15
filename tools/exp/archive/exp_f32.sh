#!/bin/bash
# A/B of launch-structure knobs on the default (fp32 arithmetic, fp16 storage) search: one line per variant
cd "$GRAFT_REPO_ROOT"
P=${PRNS:-6}
run() {
  env "$@" timeout 600 python bench.py --workload b1c --prns $P --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(tag.ljust(70), 'search', round(j['stage_ms']['search_ms'],2), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), 'frac', round(r['frac'],3), 'det', j['config']['satellites_detected'])
    elif 'amdgpu.ids' not in l: print(l.rstrip())
" "$*"
}
run BDS_X=0
run BDS_LIB_PATH=tools/variants/libbds_lean3.so
