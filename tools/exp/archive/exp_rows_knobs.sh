#!/bin/bash
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 --workload b1c --prns 8 --steps 3 --warmup 1 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(40), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3))
" "$tag"; }
for g in 12 17 21 26 34 41 51 67; do run "gchunk $g" BDS_ACQ_GCHUNK=$g; done
for g in 512 768 1024; do run "rows_grid $g" BDS_ACQ_ROWS_GRID=$g; done
run "overlap" BDS_ACQ_OVERLAP=1
