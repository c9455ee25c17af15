#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in 3 5 6 8 12 24; do
  BDS_ACQ_WCOLS_GRID=$g timeout 300 python bench.py --prns 8 --no-cpu-baseline --no-tracking --no-fast-path --steps 3 --warmup 1 > gpurun_out/wgrid_$g.json 2> gpurun_out/wgrid_$g.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/wgrid_$g.json") if l.startswith("{")][-1])
r = d["roofline"]
print("grid/CU $g: cols_ms", round(r["cols_ms"], 3), "rows_ms", round(r["rows_ms"], 3), "n_extra", r["n_extra"])
PY
done
bash tools/pmc_run.sh
grep -A45 "== k_cols_wave_f" gpurun_out/pmc_summary.txt
