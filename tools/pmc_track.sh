#!/bin/bash
# PMC counters of the tracking kernels (tools/bench_track.py, WB), one rocprofv3 --pmc pass per counter set
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmct
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d gpurun_out/pmct -o pass$i -- python tools/bench_track.py --mode ${MODE:-WB} --epochs 20 > gpurun_out/pmct_pass$i.log 2>&1
  echo "pass$i: $set rc=$?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES
SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH
SETS
python tools/pmc_summary.py gpurun_out/pmct/pass*_results.db > gpurun_out/pmct_summary_${TAG:-x}.txt 2>&1
rm -rf gpurun_out/pmct
