export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
for lt in 2 3; do echo -n "LOGT=$lt: "; BDS_ACQ_LOGT=$lt timeout 300 python bench.py --workload b1c --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ms/step', round(d['ms_per_step'],1), 'search', round(d['stage_ms']['search_ms'],1), len(d['config']['satellites_detected']))
"; done
