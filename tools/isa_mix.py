#!/usr/bin/env python3
"""Static instruction mix of the two search kernels' hot loops from the compiler's ISA (no GPU needed):
    hipcc ... -S bds_acq.hip -o acq.s ; python tools/isa_mix.py acq.s > profiles/r03_valu_bound.txt
Per kernel: VALU instructions by issue class (SIMD cycles per wave-instruction measured with tools/probe/valu_rate.hip:
2 for v_fma/add/sub/mul/mov/and/xor f32/b32, 4 for v_pk_*, v_cvt_*, v_max*, v_cmp*, v_cndmask, v_dot2*, v_alignbit, v_lshl_add,
v_mul_lo, v_perm, 8 for v_sqrt/v_rsq/v_rcp), LDS instructions (a ds_read2/ds_write2_b64 counts as two) priced three ways:
  * marginal issue cost beside vector work, per SIMD with two waves on it (tools/probe/coissue.hip, profiles/r04_coissue.txt:
    "16fma,w" 44.6 vs "16fma" 34.6 cycles per group per SIMD -> ds_write_b64 10, ds_read_b64 6.2 SIMD-cycles);
  * time of the CU's LDS unit (profiles/r03_lds_pattern.txt: 6.5 CU-cycles per ds_write_b64, 2.1 per ds_read_b64);
  * the additive round-3 figure (8 / 24 SIMD-cycles: the LDS unit's throughput expressed per SIMD), kept for comparison."""
import collections
import json
import re
import sys

CYC4 = ("v_pk_", "v_fma_mix", "v_cvt_", "v_max", "v_min", "v_cmp", "v_cndmask", "v_dot2", "v_alignbit", "v_lshl_add", "v_mul_lo", "v_perm", "v_mad_u", "v_lshlrev_b64", "v_mbcnt")
CYC8 = ("v_sqrt", "v_rsq", "v_rcp", "v_sin", "v_cos", "v_exp", "v_log")


def kernel_body(lines, pat):
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l and l.rstrip().endswith(":") or ("; @" in l and pat in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end]


def mix(body, lo, hi):
    c = collections.Counter()
    for l in body[lo:hi]:
        l = l.split(";")[0].strip()
        if not l or l.startswith("."):
            continue
        c[l.split()[0]] += 1
    valu = {2: 0, 4: 0, 8: 0}
    for op, n in c.items():
        if op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_mfma", "v_accvgpr")):
            k = 8 if op.startswith(CYC8) else 4 if op.startswith(CYC4) else 2
            valu[k] += n
    ds_r = sum(n * (2 if "read2" in op else 1) for op, n in c.items() if op.startswith("ds_read"))
    ds_w = sum(n * (2 if "write2" in op else 1) for op, n in c.items() if op.startswith("ds_write"))
    vm = sum(n for op, n in c.items() if op.startswith(("global_", "buffer_", "flat_", "scratch_")))
    return c, valu, ds_r, ds_w, vm


def report(name, body, lo, hi, items_note):
    c, valu, ds_r, ds_w, vm = mix(body, lo, hi)
    n_valu = sum(valu.values())
    cyc_valu = sum(k * v for k, v in valu.items())
    cyc_lds = 8 * ds_r + 24 * ds_w
    cyc_lds_marg = 6.2 * ds_r + 10.0 * ds_w
    cyc_lds_unit = 2.1 * ds_r + 6.5 * ds_w  # CU-cycles
    print(f"== {name}  (ISA lines {lo}..{hi}: {items_note})")
    print(f"   VALU instructions {n_valu}: {valu[2]} x 2 cycles, {valu[4]} x 4, {valu[8]} x 8  ->  {cyc_valu} SIMD cycles, {cyc_valu / n_valu:.2f} per instruction")
    print(f"   LDS  {ds_r} b64 reads + {ds_w} b64 writes: marginal issue cost {cyc_lds_marg:.0f} SIMD cycles (6.2 / 10), LDS-unit time {cyc_lds_unit:.0f} CU cycles (2.1 / 6.5), "
          f"round-3 additive figure {cyc_lds} (8 / 24);   {vm} global/buffer memory instructions")
    top = sorted(((n, op) for op, n in c.items() if op.startswith(("v_", "ds_"))), reverse=True)[:14]
    print("   " + ", ".join(f"{op} {n}" for n, op in top))
    n_mfma = sum(n for op, n in c.items() if op.startswith("v_mfma"))
    if n_mfma:
        print(f"   MFMA {n_mfma} x v_mfma_f32_16x16x32_f16 (16 matrix-pipe cycles each, 4 passes): {16 * n_mfma} cycles of the SIMD's matrix pipe")
    return dict(valu_insts=n_valu, valu_cycles=cyc_valu, cycles_per_inst=cyc_valu / n_valu, lds_cycles=cyc_lds, lds_marginal_cycles=cyc_lds_marg,
                lds_unit_cu_cycles=cyc_lds_unit, lds_reads=ds_r, lds_writes=ds_w, vmem_insts=vm, class_counts={str(k): v for k, v in valu.items()},
                mfma_insts=n_mfma, mfma_cycles=16 * n_mfma)


def main_pfa(lines, outp):
    """the N-point pair (csrc/bds_acq_pfa.h, round 6):  python tools/isa_mix.py acq.s profiles/r06_isa_mix.json pfa"""
    out = {}
    # column pass: ONE OUTPUT BLOCK of a wave item (the loop over nb: 16 real outputs = 8 lags t1, x 4 lags t3 x 12 lags t2 x 2 components;
    # seven of them per wave item, whose 24 buffer loads stand in front of the loop)
    b = kernel_body(lines, "k_pfa_colsILi2ELb0EE")
    first_mfma = next(i for i, l in enumerate(b) if "v_mfma" in l)
    hdr = [i for i, l in enumerate(b) if l.startswith(".LBB") and "Loop" in l and i < first_mfma]
    lo = max(hdr)
    # (ds_read_b128 counted as two b64 reads)
    hi = next(i for i, l in enumerate(b) if i > first_mfma and "v_readlane_b32" in l)
    body = [l.replace("ds_read_b128", "ds_read2_b64") for l in b]
    out["cols"] = report("k_pfa_cols<2, false>", body, lo, hi, "one output block of a wave item (7 per item): 8 x 4 x 12 lags x 2 components = 12 point-components per lane, bound pass")
    b = kernel_body(lines, "k_pfa_rowsILi2EE")
    bar = [i for i, l in enumerate(b) if "s_barrier" in l]
    hdr = [(i, l.split(":")[0]) for i, l in enumerate(b) if l.startswith(".LBB") and "Loop" in l and i < bar[0]]
    lo, lab = max(hdr)
    hi2 = max(i for i, l in enumerate(b) if "s_cbranch" in l and l.split()[-1] == lab)
    out["rows"] = report("k_pfa_rows<2>", b, lo, hi2, "one cell of a row pair, 2 components: 2 x 25 point-components per lane (125 of 128 lanes of a row at work)")
    if outp:
        json.dump(out, open(outp, "w"), indent=1)


def main():
    lines = open(sys.argv[1]).read().split("\n")
    if len(sys.argv) > 3 and sys.argv[3] == "pfa":
        return main_pfa(lines, sys.argv[2])
    out = {}
    # column pass: one tile (both components) per workgroup; everything up to the wave maximum is the hot part
    b = kernel_body(lines, "k_cols_wave_fILi768ELi2ELb0E7__half2Li6ELb1ELb1EE")
    hi = next(i for i, l in enumerate(b) if "v_readlane_b32" in l)
    out["cols"] = report("k_cols_wave_f<768, 2, false, __half2, 6, true, true>", b, 0, hi, "one 768 x 8 tile, 2 components = 48 point-components per lane (each through phase A and phase B)")
    # row pass: the per-cell loop (2 components of one 4096-point row on 256 threads)
    rows_kernel = ("k_rows_wave_fILi2ELb1ELb1EE", "k_rows_wave_f<2, true, true>")
    b = kernel_body(lines, rows_kernel[0])
    # the cell loop = the innermost loop that holds the barriers: from its header label to its back-edge branch
    hdr = [(i, l.split(":")[0]) for i, l in enumerate(b) if l.startswith(".LBB") and "Depth=2" in l and "Loop" in l]
    bar = [i for i, l in enumerate(b) if "s_barrier" in l]
    lo, lab = max((i, n) for i, n in hdr if i < bar[2])
    hi2 = max(i for i, l in enumerate(b) if "s_cbranch" in l and l.split()[-1] == lab)
    out["rows"] = report(rows_kernel[1], b, lo, hi2, "one cell of one row, 2 components = 32 point-components per lane")
    json.dump(out, open(sys.argv[2], "w"), indent=1) if len(sys.argv) > 2 else None


if __name__ == "__main__":
    main()
