#!/bin/bash
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for q in 1 2 4 8 16; do
  BDS_ACQ_WCOLS_QCHUNK=$q timeout 300 python bench.py --prns 8 --no-cpu-baseline --no-tracking --no-strict-f32 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('qchunk', sys.argv[1], 'pair', round(r['pair_ms'],3), 'rows', round(r['rows_ms'],3), 'cols', round(r['cols_ms'],3), 'n_extra', r['n_extra'])
" $q
done
