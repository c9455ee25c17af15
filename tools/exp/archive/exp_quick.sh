#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 $ARGS 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(40), 'ms/step', round(j['ms_per_step'],3), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), 'n_extra', r['n_extra'])
    elif 'amdgpu.ids' not in l and ('Error' in l or 'error' in l): print(l.rstrip())
" "$tag"; }
ARGS="--workload b1c --prns 8 --steps 3 --warmup 1"
run "b1c wave" A=1
run "b1c tile (round 2)" BDS_ACQ_WCOLS=0
for v in $(ls tools/variants/libbds_*.so 2>/dev/null); do run "b1c $v" BDS_LIB_PATH=$v; done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests.log | tail -2
