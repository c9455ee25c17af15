#!/bin/bash
# stage times (forward / search / refine) of both full acquisitions, in-tree library, optional env settings in $ENVS ("A=1;X=2")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
IFS=';' read -ra ES <<< "${ENVS:-A=1}"
for e in "${ES[@]}"; do for w in ${WLS:-b1c b2a}; do
  env $e python bench.py --workload $w --no-cpu-baseline --no-tracking --no-strict-f32 --no-b2a --steps 5 --warmup 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$e $w', round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['stage_ms'].items()}, 'pair', round(j['roofline']['pair_ms'],3), str(j['config']['results_sha256'])[:24])"
done; done
