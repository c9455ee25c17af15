#!/bin/bash
# round 5: does the inter-pass buffer stay in the 256 MB Infinity Cache when a launch pair carries only a few cells?
# small groups (BDS_ACQ_GROUP) x row-workgroup chunk (BDS_ACQ_GCHUNK) x the column pass on a second stream beside the next
# group's row pass (BDS_ACQ_OVERLAP).  Hooks build; prints the search time per cell.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BDS_LIB_PATH="$GRAFT_REPO_ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so"  # the tuning switches exist in the test-hooks build only
P=${PRNS:-4}
run() {
  env "$@" timeout 300 python bench.py --workload b1c --prns $P --steps 2 --warmup 1 --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --no-b2a --no-cold 2>&1 | python -c "
import sys,json
tag=sys.argv[1]; P=int(sys.argv[2])
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stage_ms']; print(tag.ljust(60), 'search', round(s['search_ms'],2), 'us/cell', round(s['search_ms']*1e3/(P*201),2), 'det', len(j['config']['satellites_detected']), 'sha', str(j['config'].get('results_sha256'))[8:20])
    elif 'amdgpu.ids' not in l and ('rror' in l or 'Traceback' in l): print(l.rstrip())
" "$*" $P
}
{
run A=1
for g in 2 3 4 6 8 12 24; do
  run BDS_ACQ_GROUP=$g
  run BDS_ACQ_GROUP=$g BDS_ACQ_OVERLAP=1
  if [ $g -ge 4 ]; then run BDS_ACQ_GROUP=$g BDS_ACQ_GCHUNK=$((g/2)) BDS_ACQ_OVERLAP=1; fi
done
run A=1
} 2>&1 | tee gpurun_out/r05_mall_groups.txt
