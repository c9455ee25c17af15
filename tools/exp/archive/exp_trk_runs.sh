#!/bin/bash
# A/B of the run-based tracking correlator: grid size (chunks per workgroup), samples per lane, vs the per-sample kernel
cd "$GRAFT_REPO_ROOT"
M=${MODE:-WB}
run() {
  env "$@" timeout 300 python tools/bench_track.py --mode $M --epochs ${EPOCHS:-200} 2>&1 | python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(tag.ljust(78), 'us/epoch', round(j['ms_per_epoch']*1e3,2), 'GB/s', round(j['int8_read_GBps'],1))
    elif 'amdgpu.ids' not in l: print(l.rstrip())
" "$*"
}
run BDS_TRK_PERSAMPLE=1
run BDS_X=0
for nb in ${NBS:-122 61 31}; do run BDS_TRK_NBLOCKS=$nb; done
run BDS_TRK_CHUNK=2048
for nb in ${NBS:-122 61 31}; do run BDS_TRK_CHUNK=2048 BDS_TRK_NBLOCKS=$nb; done
for v in ${VARIANTS:-}; do
  run BDS_LIB_PATH=tools/variants/libbds_$v.so
  for nb in ${NBS:-122 61 31}; do run BDS_LIB_PATH=tools/variants/libbds_$v.so BDS_TRK_NBLOCKS=$nb; done
done
