# cells per row-pass workgroup (BDS_ACQ_GCHUNK) x group size, full 63-PRN B1C search
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
for gc in "48 16" "48 24" "67 17" "67 23" "67 34" "67 67" "101 26" "101 34" "101 51" "201 34" "201 67"; do set -- $gc; echo -n "GROUP=$1 GCHUNK=$2: "; BDS_ACQ_GROUP=$1 BDS_ACQ_GCHUNK=$2 timeout 300 python bench.py --workload b1c --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ms/step', round(d['ms_per_step'],1), 'search', round(d['stage_ms']['search_ms'],1), len(d['config']['satellites_detected']))
"; done
