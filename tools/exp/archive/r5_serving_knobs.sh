#!/bin/bash
# round 5: the column pass's work-list knob (adjacent 128-byte lines of a cell kept together) re-checked in the serving mode. Hooks build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"
run() {
  env "$@" timeout 300 python bench.py ${LEAN:-} --workload b1c --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a --no-cold 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; r=j['roofline']; print(tag.ljust(34), 'ms/step', round(j['ms_per_step'],2), 'frac', round(r['frac'],4), 'rows', round(r['rows_ms'],3), 'cols', round(r['cols_ms'],3), 'sha', str(j['config'].get('results_sha256'))[8:20])
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*"
}
{ for rep in 1 2; do for q in ${QS:-4 2 8 16}; do run BDS_ACQ_WCOLS_QCHUNK=$q; done; done; } 2>&1 | tee gpurun_out/${OUT:-r05_serving_knobs.txt}
