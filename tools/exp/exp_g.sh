for g in 1 2 3 4 6 8 12 16; do echo -n "GROUP=$g: "; BDS_ACQ_GROUP=$g timeout 300 python bench.py --workload b1c --steps 1 --warmup 1 --no-cpu-baseline --prns 8 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('search', round(d['stage_ms']['search_ms'],1), 'us/cell', round(d['stage_ms']['search_ms']*1e3/(8*201),2))
"; done
