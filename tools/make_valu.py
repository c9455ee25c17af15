#!/usr/bin/env python3
"""profiles/valu_<workload>.json: what bench.py prints as roofline.valu -- the SIMD-issue bound of one launch pair.
    python tools/make_valu.py gpurun_out/pmc_summary.txt isa_mix.json b1c 201
Inputs: the PMC summary (tools/pmc_run.sh: SQ_INSTS_VALU, SQ_INSTS_LDS per dispatch of each kernel, hardware counts) and the
static instruction mix of the kernels' hot loops (tools/isa_mix.py: issue cycles per instruction class).  The hardware count
fixes HOW MANY vector instructions a dispatch issues; the static mix says how many SIMD cycles one of them costs on average and
how many cycles the LDS instructions add (they occupy the issuing SIMD: 8 cycles per ds_read_b64, 24 per ds_write_b64)."""
import json
import re
import sys

src, mixp, workload, cells = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
mix = json.load(open(mixp))
txt = open(src).read()
blocks = {b.split("\n")[0]: b for b in re.split(r"^== ", txt, flags=re.M) if b.strip()}


def counters(prefix):
    if prefix == "k_rows_":  # the row pass of the search: wave-private kernel when the plan has one, k_rows_inv_f otherwise
        prefix = "k_rows_wave_f" if any(n.startswith("k_rows_wave_f") for n in blocks) else "k_rows_inv_f"
    for name, b in blocks.items():
        if name.startswith(prefix):
            return name, {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+([\d.]+)", b, re.M)}
    raise SystemExit(f"no kernel {prefix} in {src}")


SIMDS, CLOCK = 256 * 4, 2.4  # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
kern = {}
tot_valu = tot_lds = insts = 0.0
for key, prefix in (("rows", "k_rows_"), ("cols", "k_cols_wave_f")):
    name, c = counters(prefix)
    m = mix[key]
    valu_cyc = c["SQ_INSTS_VALU"] * m["cycles_per_inst"] / SIMDS
    lds_cyc = c["SQ_INSTS_VALU"] / m["valu_insts"] * m["lds_cycles"] / SIMDS  # LDS issue cycles scale with the hot-loop count
    kern[key] = {"kernel": name, "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS"),
                 "static_cycles_per_valu_inst": m["cycles_per_inst"], "valu_cycles_per_simd": valu_cyc, "lds_issue_cycles_per_simd": lds_cyc,
                 "bound_ms": (valu_cyc + lds_cyc) / (CLOCK * 1e6), "measured_ms": c["~duration_ns"] / 1e6,
                 "SQ_WAIT_ANY_over_SQ_WAVE_CYCLES": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
                 "SQ_LDS_BANK_CONFLICT_over_IDX_ACTIVE": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None}
    tot_valu += valu_cyc
    tot_lds += lds_cyc
    insts += c["SQ_INSTS_VALU"]
out = {"workload": workload, "cells_per_pair": cells, "insts_per_pair": insts,
       "cycles_per_inst": sum(kern[k]["SQ_INSTS_VALU"] * kern[k]["static_cycles_per_valu_inst"] for k in kern) / insts,
       "valu_issue_cycles_per_simd": tot_valu, "lds_issue_cycles_per_simd": tot_lds, "clock_GHz": CLOCK,
       "bound_ms": (tot_valu + tot_lds) / (CLOCK * 1e6), "kernels": kern,
       "source": f"{src} (rocprofv3 --pmc, tools/pmc_run.sh) + {mixp} (tools/isa_mix.py)"}
json.dump(out, open(f"profiles/valu_{workload}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
