# A/B of the column-pass magnitude: v_dot2_f32_f16 vs convert + fma (same box, full 63 PRNs)
for rep in 1 2; do for m in 0 1; do
  touch bds-3-b1c-b2a-sdr-receiver_amd/csrc/bds_acq.hip
  BDS_HIPCC_EXTRA="-DBDS_MAG_DOT2=$m"  # (macro since removed; kept as the record of the experiment) ./build.sh 2>&1 | grep -q built || { echo "build failed: $m"; continue; }
  echo -n "DOT2=$m: "
  timeout 300 python bench.py --workload b1c --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ms/step', round(d['ms_per_step'],1), 'search', round(d['stage_ms']['search_ms'],1))
"
done; done
