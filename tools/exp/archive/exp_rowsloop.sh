# row-pass cell loop: prefetch on/off x occupancy bound x cells per workgroup (full B1C search)
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
for v in "-DBDS_ROWS_PREFETCH=1 -DBDS_ROWS_OCC=3" "-DBDS_ROWS_PREFETCH=0 -DBDS_ROWS_OCC=4" "-DBDS_ROWS_PREFETCH=0 -DBDS_ROWS_OCC=3"; do
  touch bds-3-b1c-b2a-sdr-receiver_amd/csrc/bds_acq.hip
  BDS_HIPCC_EXTRA="$v" ./build.sh 2>&1 | grep -q built || { echo "build failed: $v"; continue; }
  for gc in "16 4" "16 16" "32 8" "32 16" "48 16"; do set -- $gc; echo -n "$v GROUP=$1 GCHUNK=$2: "; BDS_ACQ_GROUP=$1 BDS_ACQ_GCHUNK=$2 timeout 300 python bench.py --workload b1c --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ms/step', round(d['ms_per_step'],1), 'search', round(d['stage_ms']['search_ms'],1), len(d['config']['satellites_detected']))
"; done; done
