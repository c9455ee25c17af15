#!/bin/bash
# forward-pass variants (BDS_FWD_T: columns per workgroup of k_cols_fwd_t) on one box: forward_ms of cfg3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" fwd8 fwd16 "" fwd8 fwd16; do
  env ${v:+BDS_LIB_PATH=tools/variants/libbds_$v.so} A=1 python bench.py --workload b1c --prns 8 --no-cpu-baseline --no-tracking --no-strict-f32 --no-b2a --steps 3 --warmup 1 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('${v:-in-tree (T=4)}'.ljust(18), 'forward_ms', round(j['stage_ms']['forward_ms'],3), 'ms/step', round(j['ms_per_step'],3), str(j['config']['results_sha256'])[:24])
    elif 'rror' in l: print(l.rstrip())"
done
