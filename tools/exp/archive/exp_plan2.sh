export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
for pl in 768x4096 1024x3072; do for lt in 2 3; do echo -n "plan=$pl LOGT=$lt: "; BDS_ACQ_LOGT=$lt BDS_ACQ_FORCE_L1L2=$pl timeout 300 python bench.py --workload b1c --steps 1 --warmup 1 --no-cpu-baseline --prns 8 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('search', round(d['stage_ms']['search_ms'],1), 'us/cell', round(d['stage_ms']['search_ms']*1e3/(8*201),2), d['config']['satellites_detected'])
"; done; done
