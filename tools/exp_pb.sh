# cells per multi-PRN launch pair on the B2a plan (BDS_ACQ_PBCELLS), with the row-pass chunk
for c in 104 208 416 832 1664; do echo -n "PBCELLS=$c: "; BDS_ACQ_PBCELLS=$c timeout 300 python bench.py --workload b2a --steps 5 --warmup 2 --no-cpu-baseline --no-tracking 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ms/step', round(d['ms_per_step'],2), 'search', round(d['stage_ms']['search_ms'],2), len(d['config']['satellites_detected']))
"; done
