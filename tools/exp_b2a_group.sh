# B2a plan (256 x 1280), fused chain: group size x cells per row workgroup
for g in 13 16 26; do for f in 1 2 4 7 13; do echo -n "B2a GROUP=$g FCHUNK=$f: "; BDS_ACQ_FCHUNK=$f BDS_ACQ_GROUP=$g timeout 300 python bench.py --workload b2a --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ms/step', round(d['ms_per_step'],2), 'search', round(d['stage_ms']['search_ms'],2), len(d['config']['satellites_detected']))
"; done; done
