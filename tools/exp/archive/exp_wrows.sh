#!/bin/bash
# same-box comparison of the wave-private row pass (bds_acq_wrows.h) against k_rows_inv_f, its chunk / grid knobs and the build
# variants under tools/variants/ (cfg3, 8 PRNs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 --workload b1c --prns 8 --steps 3 --warmup 1 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(44), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3))
" "$tag"; }
run "wave rows (default)" A=1
run "k_rows_inv_f" BDS_ACQ_WROWS=0
for v in $(ls tools/variants/libbds_*.so 2>/dev/null); do run "$v" BDS_ACQ_NO_SELFCHECK=1 BDS_LIB_PATH=$v; done
for g in 17 26 41 51 67 101 201; do run "gchunk $g" BDS_ACQ_GCHUNK=$g; done
for g in 512 1024; do run "rows_grid $g" BDS_ACQ_ROWS_GRID=$g; done
run "wave rows (default) again" A=1
