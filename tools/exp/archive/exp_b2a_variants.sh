#!/bin/bash
# same-box A/B on cfg2 (B2a, 63 PRNs x 26 bins, plan 256 x 1280) and on the fp32-storage cfg3 search (8 PRNs): in-tree library
# against every build variant under tools/variants/, alternating twice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 $ARGS 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(44), 'ms/step', round(j['ms_per_step'],3), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3))
" "$tag"; }
for rep in 1 2; do
  ARGS="--workload b2a --steps 10 --warmup 2"
  run "b2a in-tree" A=1
  for v in $(ls tools/variants/libbds_*.so 2>/dev/null); do run "b2a $v" BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v; done
  ARGS="--workload b1c --prns 8 --steps 3 --warmup 1"
  run "b1c fp32 storage in-tree" BDS_ACQ_FP16=0
  for v in $(ls tools/variants/libbds_*.so 2>/dev/null); do run "b1c fp32 storage $v" BDS_ACQ_FP16=0 BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v; done
done
