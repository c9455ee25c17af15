#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_acq_gpu.py tests/test_sieve_gpu.py -x -q 2>&1 | tail -2
for g in 0 2 6; do
  BDS_ACQ_WCOLS_GRID=$g timeout 300 python bench.py --prns 8 --no-cpu-baseline --no-tracking --no-fast-path --steps 3 --warmup 1 > gpurun_out/wgrid_$g.json 2> gpurun_out/wgrid_$g.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/wgrid_$g.json") if l.startswith("{")][-1])
r = d["roofline"]
print("grid/CU $g: cols_ms", round(r["cols_ms"], 3), "rows_ms", round(r["rows_ms"], 3), "n_extra", r["n_extra"])
PY
done
ARGS="--workload b1c --steps 1 --warmup 0 --no-cpu-baseline --no-tracking --no-fast-path --prns 2"
timeout 600 rocprofv3 --pmc FETCH_SIZE TCC_HIT TCC_MISS -d gpurun_out/pmc -o p1 -- python bench.py $ARGS > gpurun_out/pmc_p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc -o p2 -- python bench.py $ARGS > gpurun_out/pmc_p2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc/p*_results.db > gpurun_out/pmc_summary2.txt 2>&1; rm -rf gpurun_out/pmc
grep -A14 "== k_cols_wave_f" gpurun_out/pmc_summary2.txt
