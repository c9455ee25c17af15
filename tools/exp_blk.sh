#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-fast-path --workload b1c --prns 8 --steps 3 --warmup 1 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(40), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), j['config']['satellites_detected'])
    elif 'rror' in l: print(l.rstrip())
" "$tag"; }
run "row-major q4" BDS_ACQ_WCOLS_ROWMAJOR=1
for q in 1 2 4 8 16; do run "blocked q$q" BDS_ACQ_WCOLS_QCHUNK=$q; done
timeout 900 python -m pytest tests/test_acq_gpu.py tests/test_sieve_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
