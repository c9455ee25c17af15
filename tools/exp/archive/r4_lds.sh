#!/bin/bash
# LDS counters (bank conflicts, FIFO stalls) of the two search kernels: in-tree library and the variants named in $VARIANTS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--workload ${WL:-b1c} --steps 1 --warmup 0 --no-cpu-baseline --no-tracking --no-strict-f32 --no-b2a --prns 2"
one() { tag=$1; shift
  rm -rf gpurun_out/pmc_lds
  env "$@" timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES \
      -d gpurun_out/pmc_lds -o p -- python bench.py $ARGS > /dev/null 2>&1
  echo "#### $tag"; python tools/pmc_summary.py gpurun_out/pmc_lds/p_results.db 2>&1 | grep -A12 "^== .*\(k_rows_wave_f\|k_cols_wave_f\|k_cols_small_f\)" | cut -c1-110
  rm -rf gpurun_out/pmc_lds; }
{ one in-tree A=1; for v in ${VARIANTS:-base}; do one $v BDS_LIB_PATH=tools/variants/libbds_$v.so; done; } 2>&1 | tee gpurun_out/r04_lds.txt
