#!/bin/bash
# Round-5 measurement pass (GPU box): whole GPU suite, default bench lines, kernel-trace stats (cfg3, cfg2, tracking), PMC counters.
# Everything lands under gpurun_out/; tools/r5_collect.sh copies the summaries into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "${TESTS:-1}" = 1 ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5_gpu_tests.log 2>&1; tail -4 gpurun_out/r5_gpu_tests.log
fi
timeout 1200 python bench.py > gpurun_out/bench_b1c.json 2> gpurun_out/bench_b1c.err; echo "bench b1c rc=$?"
timeout 600 python bench.py --workload b2a > gpurun_out/bench_b2a.json 2> gpurun_out/bench_b2a.err; echo "bench b2a rc=$?"
bash tools/profile_run.sh > gpurun_out/profile_run.log 2>&1; tail -3 gpurun_out/profile_run.log
bash tools/profile_track.sh > gpurun_out/profile_track.log 2>&1; tail -3 gpurun_out/profile_track.log
bash tools/pmc_run.sh > gpurun_out/pmc_run.log 2>&1; tail -2 gpurun_out/pmc_run.log
cp gpurun_out/pmc_summary.txt gpurun_out/pmc_summary_default.txt
python - <<'PY'
import json
for w in ("b1c", "b2a"):
    try:
        j = json.loads(open("gpurun_out/bench_%s.json" % w).read().strip().splitlines()[-1]); r = j["roofline"]
        print(w, "ms/step %.3f frac %.3f pair %s stage %s" % (j["ms_per_step"], r["frac"], r.get("pair_ms"), j.get("stage_ms")))
        for k in ("b2a", "cold", "tracking", "tracking_full"):
            if k in j: print("  ", k, json.dumps(j[k])[:600])
    except Exception as e:
        print(w, "unreadable", e)
PY
