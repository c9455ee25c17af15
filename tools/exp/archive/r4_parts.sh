#!/bin/bash
# where the pair's time goes: in-tree library against timing variants whose results are INVALID (tools/build_variant.sh):
#   nostore -DBDS_EXP_ROWS_NOSTORE  row pass without its stores to the inter-pass buffer (round 4: rows 1.62 vs 1.72 ms)
# (-DBDS_EXP_WC_NOLOAD, the column pass without its reads, is no use with the candidate-list protocol: an all-zero surface ties
#  everywhere, the list runs over and the call falls back to the run-time-plan kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in "" $(ls tools/variants/libbds_*.so 2>/dev/null | grep -v phases); do
  env BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v timeout 300 python bench.py --workload b1c --prns ${PRNS:-6} --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-b2a 2>&1 | python -c "
import sys,json
tag=(sys.argv[1] if len(sys.argv)>1 and sys.argv[1] else 'in-tree')
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(tag.ljust(44), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3))
    elif 'amdgpu.ids' not in l and 'Error' in l: print(l.rstrip())
" "$v"
done; done 2>&1 | tee gpurun_out/r04_parts.txt
