#!/bin/bash
# PMC counters of the two row-pass kernels side by side (wave-private vs k_rows_inv_f), two passes each
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--workload b1c --steps 1 --warmup 0 --no-cpu-baseline --no-tracking --no-strict-f32 --prns 2"
for mode in 1 0; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES"; do
    i=$((i+1))
    BDS_ACQ_WROWS=$mode timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmcr$mode -o pass$i -- python bench.py $ARGS > gpurun_out/pmcr${mode}_pass$i.log 2>&1; echo "mode $mode pass$i rc=$?"
  done
  python tools/pmc_summary.py gpurun_out/pmcr$mode/pass*_results.db > gpurun_out/pmc_rows_$mode.txt 2>&1; rm -rf gpurun_out/pmcr$mode
  grep -A18 "== k_rows" gpurun_out/pmc_rows_$mode.txt | head -20
done
