#!/bin/bash
# copy the summaries of a tools/r6_profile.sh pass from gpurun_out/ into profiles/ (run here, after the gpurun call) and rebuild
# the derived files (traffic_b1c.json, valu_b1c.json) for the N-point pair
cd "$(dirname "${BASH_SOURCE[0]}")/.."
cp gpurun_out/kernel_stats_b1c.txt profiles/r06_b1c_kernel_stats.txt
cp gpurun_out/kernel_stats_b2a.txt profiles/r06_b2a_kernel_stats.txt
cp gpurun_out/kernel_stats_trk_B2A.txt profiles/r06_trk_b2a_kernel_stats.txt
cp gpurun_out/kernel_stats_trk_WB.txt profiles/r06_trk_wb_kernel_stats.txt
tail -1 gpurun_out/bench_b1c.json > profiles/r06_bench_b1c.json
tail -1 gpurun_out/bench_b2a.json > profiles/r06_bench_b2a.json
cp gpurun_out/bench_under_rocprof_b1c.json profiles/r06_bench_b1c_under_rocprof.json 2>/dev/null
python tools/make_traffic.py gpurun_out/pmc_summary_default.txt b1c ${PMC_CELLS:-1608} profiles/r06_b1c_pmc.txt k_pfa_cols "round 6" 2 > /dev/null
python tools/make_valu.py gpurun_out/pmc_summary_default.txt profiles/r06_isa_mix.json b1c ${PMC_CELLS:-1608} "round 6" pfa > /dev/null
python - <<'PY'
import json
j = json.load(open("profiles/r06_bench_b1c.json")); r = j["roofline"]; v = r.get("valu") or {}
print("b1c: ms/step %.1f frac %.3f pair %.3f rows %.3f cols %.3f clock %s" % (j["ms_per_step"], r["frac"], r["pair_ms"], r["rows_ms"], r["cols_ms"], v.get("shader_clock_GHz")))
print("b2a key:", j.get("b2a", {}).get("ms_per_step"), (j.get("b2a") or {}).get("stage_ms"))
t = json.load(open("profiles/traffic_b1c.json")); print("traffic GB/pair %.2f (%.1f MB per cell)" % (t["bytes_per_pair"] / 1e9, t["per_cell_MB"]))
u = json.load(open("profiles/valu_b1c.json")); print("bound ms %.3f" % u["bound_ms"], {k: round(x["valu_busy"] or 0, 3) for k, x in u["kernels"].items()})
PY
