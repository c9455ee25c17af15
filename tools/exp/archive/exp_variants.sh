#!/bin/bash
# same-box A/B of the in-tree library against every build variant under tools/variants/ (cfg3, 8 PRNs), alternating twice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 --workload b1c --prns 8 --steps 3 --warmup 1 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(44), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3))
" "$tag"; }
for rep in 1 2; do
  run "in-tree" A=1
  for v in $(ls tools/variants/libbds_*.so 2>/dev/null); do run "$v" BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v; done
done
