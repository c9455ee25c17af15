#!/usr/bin/env python3
"""profiles/valu_<workload>.json: what bench.py prints as roofline.valu -- issue model of one launch pair of the search.
    python tools/make_valu.py gpurun_out/pmc_summary.txt profiles/r04_isa_mix.json b1c 201 "round 4"
Inputs: the PMC summary (tools/pmc_run.sh: hardware counts per dispatch of each kernel) and the static instruction mix of the
kernels' hot loops (tools/isa_mix.py).  The hardware count fixes HOW MANY vector instructions a dispatch issues; the static mix
says how many pipe cycles one of them takes on average (2 for a plain fp32 instruction, 4 for a packed one / a conversion) and
how the LDS instructions split into reads and writes.

Round 4 replaces the additive bound of round 3 (VALU cycles + 8 / 24 SIMD-cycles per LDS read / write: the latter are the LDS
UNIT's throughput expressed per SIMD, not issue-slot occupancy) by what tools/probe/coissue.hip measures:
    bound = max( vector pipe cycles + marginal LDS issue cost (6.2 / 10 SIMD-cycles per ds_read_b64 / ds_write_b64 beside vector
                 work), LDS-unit time (2.1 / 6.5 CU-cycles) ) / clock
and prints the additive figure beside it.  It also records that the chip is POWER-limited under this load (1325 W of the 1400 W
cap, tools/exp/r4_power.sh): wall time per instruction does not improve with a better schedule, only with less switching."""
import json
import re
import sys

src, mixp, workload, cells = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
rnd = sys.argv[5] if len(sys.argv) > 5 else "round 4"
mix = json.load(open(mixp))
txt = open(src).read()
blocks = {b.split("\n")[0]: b for b in re.split(r"^== ", txt, flags=re.M) if b.strip()}


def counters(prefix):
    for name, b in blocks.items():
        if name.startswith(prefix):
            return name, {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+([\d.]+)", b, re.M)}
    raise SystemExit(f"no kernel {prefix} in {src}")


CUS, SIMDS, CLOCK = 256, 256 * 4, 2.4  # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
kern = {}
tot = dict(valu=0.0, marg=0.0, unit=0.0, add=0.0, insts=0.0)
PFA = len(sys.argv) > 6 and sys.argv[6] == "pfa"  # the N-point pair of round 6 (csrc/bds_acq_pfa.h)
for key, prefix in ((("rows", "k_pfa_rows"), ("cols", "k_pfa_cols")) if PFA else (("rows", "k_rows_wave_f"), ("cols", "k_cols_wave_f"))):
    name, c = counters(prefix)
    m = mix[key]
    items = c["SQ_INSTS_VALU"] / m["valu_insts"]  # hot-loop items (wave-tiles / wave-cells) of one dispatch
    valu_cyc = c["SQ_INSTS_VALU"] * m["cycles_per_inst"] / SIMDS
    marg = items * m["lds_marginal_cycles"] / SIMDS
    unit = items * m["lds_unit_cu_cycles"] / CUS
    add = items * m["lds_cycles"] / SIMDS
    elapsed = c["GRBM_GUI_ACTIVE"] / 8.0 if c.get("GRBM_GUI_ACTIVE") else None  # shader cycles of the dispatch (8 XCDs summed)
    wc = c.get("SQ_WAVE_CYCLES")
    ia = c.get("SQ_LDS_IDX_ACTIVE")
    kern[key] = {"kernel": name, "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS"),
                 "static_pipe_cycles_per_valu_inst": m["cycles_per_inst"], "valu_insts_per_item": m["valu_insts"],
                 "lds_reads_per_item": m["lds_reads"], "lds_writes_per_item": m["lds_writes"],
                 "valu_pipe_cycles_per_simd": valu_cyc, "lds_marginal_issue_cycles_per_simd": marg, "lds_unit_cycles_per_cu": unit,
                 "mfma_insts_per_item": m.get("mfma_insts", 0), "matrix_pipe_cycles_per_simd": items * m.get("mfma_cycles", 0) / SIMDS,
                 "bound_ms": max(valu_cyc + marg, unit, items * m.get("mfma_cycles", 0) / SIMDS) / (CLOCK * 1e6), "additive_r3_bound_ms": (valu_cyc + add) / (CLOCK * 1e6),
                 "measured_ms": c["~duration_ns"] / 1e6,
                 "valu_busy": valu_cyc / elapsed if elapsed else None,
                 "shader_clock_GHz_under_pmc": elapsed / c["~duration_ns"] if elapsed else None,
                 "SQ_WAIT_ANY_over_SQ_WAVE_CYCLES": c["SQ_WAIT_ANY"] / wc if wc and c.get("SQ_WAIT_ANY") else None,
                 "SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES": c["SQ_WAIT_INST_ANY"] / wc if wc and c.get("SQ_WAIT_INST_ANY") else None,
                 "SQ_LDS_DATA_FIFO_FULL_over_IDX_ACTIVE": c["SQ_LDS_DATA_FIFO_FULL"] / ia if ia and c.get("SQ_LDS_DATA_FIFO_FULL") else None,
                 "SQ_LDS_BANK_CONFLICT_over_IDX_ACTIVE": c["SQ_LDS_BANK_CONFLICT"] / ia if ia and c.get("SQ_LDS_BANK_CONFLICT") is not None else None}
    tot["valu"] += valu_cyc
    tot["marg"] += marg
    tot["unit"] += unit
    tot["add"] += add
    tot["insts"] += c["SQ_INSTS_VALU"]
power = {"measured_W": 1325, "cap_W": 1400, "sclk_MHz": 1996,
         "note": "rocm-smi sample while the cfg3 search ran (tools/exp/archive/r4_power.sh): the chip sits at its power cap, so the "
                 "clock (and with it wall time per instruction) follows the switching activity -- tools/probe/coissue.hip: a pure "
                 "v_fma_f32 stream reaches 0.96 wave-instructions per ns and SIMD (81 % of the 2.4 GHz peak) at ANY occupancy >= 2"}
if PFA:  # the round-6 pair: sustained 20-s loops, tools/power_sustained.py -> profiles/r06_power_sustained.txt (first line)
    try:
        pjs = [json.loads(l) for l in open("profiles/r06_power_sustained.txt") if l.startswith("{")]
        pj = pjs[0]
        desc = "; ".join("%s %.1f ms per call at %.0f W / %.2f GHz = %.0f J" % (q["label"], q["ms_per_call"], q["power_W"]["median"], q["sclk_MHz"]["median"] / 1e3,
                                                                                q["ms_per_call"] * q["power_W"]["median"] / 1e3) for q in pjs)
        power = {"measured_W": pj["power_W"]["median"], "cap_W": 1400, "sclk_MHz": pj["sclk_MHz"]["median"], "ms_per_call": pj["ms_per_call"],
                 "note": "tools/power_sustained.py: 20-s loops of bds_acq_run, sysfs power / clock at ~20 Hz (profiles/r06_power_sustained.txt: " + desc + ")"}
    except Exception:
        pass
out = {"workload": workload, "cells_per_pair": cells, "round": rnd, "insts_per_pair": tot["insts"],
       "valu_pipe_cycles_per_simd": tot["valu"], "lds_marginal_issue_cycles_per_simd": tot["marg"], "lds_unit_cycles_per_cu": tot["unit"],
       "clock_GHz": CLOCK,
       "bound_ms": sum(k["bound_ms"] for k in kern.values()),
       "additive_r3_bound_ms": sum(k["additive_r3_bound_ms"] for k in kern.values()),
       "model": "per kernel max(vector pipe cycles + marginal LDS issue cost, LDS-unit time, matrix-pipe cycles) / 2.4 GHz (tools/probe/coissue.hip); "
                "additive_r3_bound_ms = the round-3 sum with 8 / 24 SIMD-cycles per LDS read / write",
       "power": power,
       "kernels": kern,
       "source": f"{src} (rocprofv3 --pmc, tools/pmc_run.sh) + {mixp} (tools/isa_mix.py)"}
json.dump(out, open(f"profiles/valu_{workload}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
