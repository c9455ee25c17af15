run() { echo "== $*"; env "$@" timeout 300 python bench.py --workload b1c --steps 1 --warmup 1 --no-cpu-baseline --prns 6 2>&1 | grep -E "^\{|plan" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ms/step', round(d['ms_per_step'],1), 'search', round(d['stage_ms']['search_ms'],1), 'us/cell', round(d['stage_ms']['search_ms']*1e3/(6*201),1), 'det', d['config']['satellites_detected'])
    else: print('  ', l.strip()[:170])
"; }
export BDS_VERBOSE=1
run BDS_ACQ_LOGT=3
run BDS_ACQ_LOGT=2
run BDS_ACQ_LOGT=1
run BDS_ACQ_LOGT=2 BDS_ACQ_FORCE_L1L2=1024x3072
run BDS_ACQ_LOGT=2 BDS_ACQ_FORCE_L1L2=512x6144
