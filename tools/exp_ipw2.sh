#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" $(ls tools/variants/libbds_*.so); do for ipw in 1 2; do
  BDS_LIB_PATH=$v BDS_ACQ_WCOLS_IPW=$ipw timeout 300 python bench.py --prns 8 --no-cpu-baseline --no-tracking --no-fast-path --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print((sys.argv[2] or 'in-tree').ljust(36), 'ipw', sys.argv[1], 'pair', round(r['pair_ms'],3), 'rows', round(r['rows_ms'],3), 'cols', round(r['cols_ms'],3))
" $ipw "$v"
done; done
ARGS="--workload b1c --steps 1 --warmup 0 --no-cpu-baseline --no-tracking --no-fast-path --prns 2"
for ipw in 1 2; do
BDS_ACQ_WCOLS_IPW=$ipw timeout 600 rocprofv3 --pmc FETCH_SIZE TCC_HIT TCC_MISS -d gpurun_out/pmc -o p$ipw -- python bench.py $ARGS > gpurun_out/pmc_p$ipw.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc/p${ipw}_results.db 2>&1 | grep -A5 "== k_cols_wave_f"
done
rm -rf gpurun_out/pmc
