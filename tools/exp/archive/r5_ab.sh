#!/bin/bash
# A/B on one box: the in-tree library against tools/variants/libbds_*.so named in $VARIANTS (default: base), alternating twice;
# extra environment settings of the in-tree runs in $ENVS ("A=1;BDS_ACQ_ILV=0");
# then (TESTS=1) the acquisition parity tests on the in-tree library.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-strict-f32 --no-tracking-full --workload ${WL:-b1c} --prns ${PRNS:-8} --steps 3 --warmup 1 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(sys.argv[1].ljust(28), 'ms/step', round(j['ms_per_step'],3), 'pair', round(r['pair_ms'],3), 'rows', round(r.get('rows_ms') or 0,3), 'cols', round(r.get('cols_ms') or 0,3), 'n_extra', r['n_extra'], 'sha', str(j['config'].get('results_sha256'))[:22])
    elif 'amdgpu.ids' not in l and ('Error' in l or 'error' in l or 'Traceback' in l): print(l.rstrip())
" "$tag"; }
IFS=';' read -ra ES <<< "${ENVS:-A=1}"
for rep in 1 2; do
  for e in "${ES[@]}"; do run "in-tree $e" $e; done
  for v in ${VARIANTS:-base}; do run "$v" BDS_LIB_PATH=tools/variants/libbds_$v.so; done
done 2>&1 | tee gpurun_out/${OUT:-r05_ab.txt}
if [ "${TESTS:-0}" = 1 ]; then
  timeout 1500 python -m pytest tests/test_acq_gpu.py tests/test_sieve_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -5
fi
if [ "${FULL:-0}" = 1 ]; then
  timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -5
fi
if [ "${PHASES:-0}" = 1 ]; then
  BDS_LIB_PATH=tools/variants/libbds_phases.so timeout 300 python tools/phases.py --prns 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_phases.txt
fi
