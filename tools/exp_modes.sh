#!/bin/bash
# bench line per search arithmetic / storage mode (few PRNs): default, fp32 arithmetic on fp16 storage, pure fp32
cd "$GRAFT_REPO_ROOT"
P=${PRNS:-6}
for m in "" "BDS_ACQ_HMATH=0" "BDS_ACQ_FP16=0"; do
  echo "== mode: ${m:-default}"
  env $m timeout 600 python bench.py --workload b1c --prns $P --steps 2 --warmup 1 --no-cpu-baseline --no-tracking 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['dtype'], 'ms/step', round(j['ms_per_step'],2), 'stage', {k:round(v,2) for k,v in j['stage_ms'].items()}, 'pair_ms', round(r['pair_ms'],3), 'frac', round(r['frac'],3), 'det', j['config']['satellites_detected'])
    else: print(l.rstrip())
"
done
