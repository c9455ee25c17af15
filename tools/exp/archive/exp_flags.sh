# build variants of bds_acq.hip with extra flags and time the B1C search (8 PRNs)
for extra in "" "-fno-slp-vectorize" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" ; do
  touch bds-3-b1c-b2a-sdr-receiver_amd/csrc/bds_acq.hip
  BDS_HIPCC_EXTRA="$extra" ./build.sh > /dev/null 2>&1 || { echo "build failed: $extra"; continue; }
  echo "== extra='$extra'"
  timeout 300 python bench.py --workload b1c --steps 1 --warmup 1 --no-cpu-baseline --prns 8 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   search', round(d['stage_ms']['search_ms'],1), 'pair_us', round(d['roofline']['pair_ms']*1e3,1), 'det', d['config']['satellites_detected'])
"
done
