#!/bin/bash
# same-box A/B of tracking builds: rocprofv3 kernel time of k_trk_correlate per variant (VARIANTS = names under tools/variants)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=${MODE:-WB}
one() {
  rm -rf gpurun_out/pt
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/pt -o t -- python tools/bench_track.py --mode $M --epochs ${EPOCHS:-200} > gpurun_out/pt.log 2>&1
  echo "== $* : $(grep -o '"ms_per_epoch": [0-9.]*' gpurun_out/pt.log)"
  python tools/rocprof_summary.py $(find gpurun_out/pt -name "*_results.db" | head -1) | sed -n 3,4p | cut -c1-100
  rm -rf gpurun_out/pt
}
one BDS_TRK_PERSAMPLE=1
one BDS_X=0
for v in ${VARIANTS:-}; do one BDS_LIB_PATH=tools/variants/libbds_$v.so; done
one BDS_X=0
