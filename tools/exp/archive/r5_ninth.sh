#!/bin/bash
# round 5, ninth GPU trip: whole GPU suite on the new refinement / load code, default bench lines, the two-stream overlap A/B of the pair
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r5_gpu_tests.log 2>&1; tail -4 gpurun_out/r5_gpu_tests.log
timeout 1200 python bench.py > gpurun_out/bench_b1c.json 2> gpurun_out/bench_b1c.err; echo "bench b1c rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_b1c.json").read().strip().splitlines()[-1]); r = j["roofline"]
print("b1c ms/step %.3f frac %.3f pair %s stage %s" % (j["ms_per_step"], r["frac"], r.get("pair_ms"), j.get("stage_ms")))
for k in ("b2a", "cold", "tracking"):
    if k in j: print("  ", k, json.dumps(j[k])[:700])
PY
ENVS="A=1;BDS_ACQ_OVERLAP=1" VARIANTS="" OUT=r05_overlap_ab.txt PRNS=8 bash tools/exp/r5_ab.sh
