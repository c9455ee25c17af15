#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ipw in 1 2 3 4 8; do
  BDS_ACQ_WCOLS_IPW=$ipw timeout 300 python bench.py --prns 8 --no-cpu-baseline --no-tracking --no-fast-path --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('ipw', sys.argv[1], 'pair', round(r['pair_ms'],3), 'rows', round(r['rows_ms'],3), 'cols', round(r['cols_ms'],3), 'n_extra', r['n_extra'])
" $ipw
done
timeout 600 python -m pytest tests/test_acq_gpu.py tests/test_sieve_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -2
