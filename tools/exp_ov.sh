for ov in 0 1; do for g in 8 16; do echo -n "OVERLAP=$ov GROUP=$g: "; if [ $ov = 1 ]; then export BDS_ACQ_OVERLAP=1; else unset BDS_ACQ_OVERLAP; fi; BDS_ACQ_GROUP=$g timeout 300 python bench.py --workload b1c --steps 1 --warmup 1 --no-cpu-baseline --prns 8 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('search', round(d['stage_ms']['search_ms'],1), 'us/cell', round(d['stage_ms']['search_ms']*1e3/(8*201),2), d['config']['satellites_detected'])
"; done; done
