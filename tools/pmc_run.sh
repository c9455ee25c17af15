#!/bin/bash
# Collect PMC counters for a short B1C bench run, one rocprofv3 pass per counter set
# (--pmc only: never combined with trace domains, see the gpurun rules).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="${BENCH_ARGS:---workload b1c --steps 1 --warmup 0 --no-cpu-baseline --prns 2}"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d gpurun_out/pmc -o pass$i -- python bench.py $ARGS > gpurun_out/pmc_pass$i.log 2>&1
  echo "pass$i: $set rc=$?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT TCC_MISS
TA_TA_BUSY TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES
SETS
ls -la gpurun_out/pmc
