#!/bin/bash
# Round 3, first GPU pass of the wave-private column pass: parity suite, then same-box A/B against the round-2 tile kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gpu_tests.log
for v in new old; do
  [ $v = old ] && export BDS_ACQ_WCOLS=0 || unset BDS_ACQ_WCOLS
  timeout 600 python bench.py --no-cpu-baseline --no-tracking --no-fast-path --steps 5 --warmup 1 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  echo "$v rc=$?"; python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/ab_$v.json") if l.startswith("{")][-1])
    print("$v", d["ms_per_step"], d["roofline"])
except Exception as e:
    print("$v failed", e)
PY
done
unset BDS_ACQ_WCOLS
timeout 600 python bench.py --workload b2a --no-cpu-baseline --no-tracking --no-fast-path --steps 10 --warmup 2 > gpurun_out/ab_b2a_new.json 2>&1; tail -1 gpurun_out/ab_b2a_new.json | cut -c1-400
BDS_ACQ_WCOLS=0 timeout 600 python bench.py --workload b2a --no-cpu-baseline --no-tracking --no-fast-path --steps 10 --warmup 2 > gpurun_out/ab_b2a_old.json 2>&1; tail -1 gpurun_out/ab_b2a_old.json | cut -c1-400
bash tools/profile_run.sh
