#!/bin/bash
# timing of library variants with parts of the search kernels removed (tools/build_variant.sh; results invalid)
cd "$GRAFT_REPO_ROOT"
P=${PRNS:-6}
for v in "" $(ls tools/variants/libbds_*.so 2>/dev/null); do
  env BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v timeout 600 python bench.py --workload b1c --prns $P --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 2>&1 | python -c "
import sys,json
tag=sys.argv[1] if len(sys.argv)>1 else 'in-tree'
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(tag.ljust(44), 'search', round(j['stage_ms']['search_ms'],2), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3))
    elif 'amdgpu.ids' not in l and 'Error' in l: print(l.rstrip())
" $v
done
