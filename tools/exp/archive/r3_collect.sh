#!/bin/bash
# copy the summaries of a tools/exp/r3_profile.sh pass from gpurun_out/ into profiles/ (run here, after the gpurun call) and rebuild
# the derived files (traffic_b1c.json, valu_b1c.json)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
cp gpurun_out/kernel_stats_b1c.txt profiles/r03_b1c_kernel_stats.txt
cp gpurun_out/kernel_stats_b2a.txt profiles/r03_b2a_kernel_stats.txt
cp gpurun_out/kernel_stats_trk_B2A.txt profiles/r03_trk_b2a_kernel_stats.txt
cp gpurun_out/kernel_stats_trk_WB.txt profiles/r03_trk_wb_kernel_stats.txt
cp gpurun_out/pmc_summary_f32.txt profiles/r03_b1c_f32_pmc.txt
cp gpurun_out/sieve_error_cfg3.txt profiles/r03_sieve_error.txt
tail -1 gpurun_out/bench_b1c.json > profiles/r03_bench_b1c.json
tail -1 gpurun_out/bench_b2a.json > profiles/r03_bench_b2a.json
cp gpurun_out/bench_under_rocprof_b1c.json profiles/r03_bench_b1c_under_rocprof.json
python tools/make_traffic.py gpurun_out/pmc_summary_default.txt b1c 201 profiles/r03_b1c_pmc.txt > /dev/null
python tools/make_valu.py gpurun_out/pmc_summary_default.txt profiles/r03_isa_mix.json b1c 201 > /dev/null
python - <<'PY'
import json
j = json.load(open("profiles/r03_bench_b1c.json")); r = j["roofline"]; v = r.get("valu") or {}
print("b1c: ms/step %.1f frac %.3f pair %.3f rows %.3f cols %.3f clock %s frac@clock %s" % (j["ms_per_step"], r["frac"], r["pair_ms"], r["rows_ms"], r["cols_ms"], v.get("shader_clock_GHz"), v.get("frac_of_issue_bound_at_shader_clock")))
t = json.load(open("profiles/traffic_b1c.json")); print("traffic GB/pair %.2f" % (t["bytes_per_pair"] / 1e9))
u = json.load(open("profiles/valu_b1c.json")); print("issue bound ms %.3f (%s)" % (u["bound_ms"], u["kernels"]["rows"]["kernel"]))
PY
