# B1C plan: several PRNs' Doppler rows per launch pair (cells = PRNs x 201)
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
for c in 201 402 804; do echo -n "B1C PBCELLS=$c: "; BDS_ACQ_MULTI_ANY=1 BDS_ACQ_PBCAP_GB=64 BDS_ACQ_PBCELLS=$c timeout 300 python bench.py --workload b1c --steps 2 --warmup 1 --no-cpu-baseline --no-tracking 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ms/step', round(d['ms_per_step'],2), 'search', round(d['stage_ms']['search_ms'],2), len(d['config']['satellites_detected']))
"; done
echo -n "B1C default: "; timeout 300 python bench.py --workload b1c --steps 2 --warmup 1 --no-cpu-baseline --no-tracking 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ms/step', round(d['ms_per_step'],2), 'search', round(d['stage_ms']['search_ms'],2))
"
