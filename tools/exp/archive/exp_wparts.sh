#!/bin/bash
# wave-private column pass: in-tree library and the BDS_EXP_WC_* variants (tools/build_variant.sh), one item per workgroup
# (default) and a persistent grid (BDS_ACQ_WCOLS_GRID=-1); results of the variants are invalid, only the times count
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=${PRNS:-6}
for grid in 0 -1; do
for v in "" $(ls tools/variants/libbds_*.so 2>/dev/null); do
  env BDS_ACQ_WCOLS_GRID=$grid BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v timeout 300 python bench.py --workload b1c --prns $P --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 2>&1 | python -c "
import sys,json
tag=(sys.argv[1] if len(sys.argv)>1 and sys.argv[1] else 'in-tree')+' grid '+sys.argv[2]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(tag.ljust(52), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), 'n_extra', r['n_extra'])
    elif 'amdgpu.ids' not in l and 'Error' in l: print(l.rstrip())
" "$v" $grid
done
done
