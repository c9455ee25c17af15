#!/bin/bash
# Collect PMC counters for a short B1C bench run, one rocprofv3 pass per counter set
# (--pmc only: never combined with trace domains, see the gpurun rules).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (--prns 32: the headline's serving mode carries 32 PRNs' Doppler rows per launch pair -- 6432 cells --, and what the row workgroups
#  of the PRNs share in L2 is part of the traffic figure; PMC_CELLS of tools/r5_collect.sh must say the same)
ARGS="${BENCH_ARGS:---workload b1c --steps 1 --warmup 0 --no-cpu-baseline --no-tracking --no-strict-f32 --no-b2a --no-cold --prns 32}"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmc -o pass$i -- python bench.py $ARGS > gpurun_out/pmc_pass$i.log 2>&1
  echo "pass$i: $set rc=$?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES
SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH
SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVE_CYCLES
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT TCC_MISS
TA_TA_BUSY TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES
SETS
python tools/pmc_summary.py gpurun_out/pmc/pass*_results.db > gpurun_out/pmc_summary.txt 2>&1
rm -rf gpurun_out/pmc
grep -c "^==" gpurun_out/pmc_summary.txt
