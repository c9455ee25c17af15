"""Phase-clock profile of the two wave-private search kernels (timing build -DBDS_EXP_PHASES, tools/build_variant.sh phases):
where a wave's life goes -- issue time of each arithmetic phase against the waits for HBM rows, LDS exchanges and workgroup
barriers.  Run on the GPU box:  BDS_LIB_PATH=tools/variants/libbds_phases.so python tools/phases.py [--prns 4]
Prints shader-clock cycles per wave and tile (column pass) / per wave and transform (row pass)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

COLS = ["rows + constants arrive (vmcnt 0)", "c0 phase A arithmetic", "c0 (no barrier)", "c0 phase A writes landed", "c0 barrier",
        "c0 phase B (3 slots: 6 exchanges + 6 butterflies)", "(unused)", "c1 phase A arithmetic", "c1 barrier (regions free)",
        "c1 phase A writes landed", "c1 barrier", "c1 phase B", "(unused)", "tail", "(unused)", "set-up: rest (phase-B constants requested, lag bases)", "set-up: kernel arguments read, item decoded", "set-up: component-0 rows requested (12 loads)", "set-up: component-1 rows requested (12 loads)", "set-up: bounds + 11 phase-A constants requested"]
ROWS = ["products + 1a butterfly, exchange issued", "exchange 1 landed (lgkmcnt 0)", "1b butterfly", "barrier 1", "exchange 2 writes landed",
        "barrier 2", "exchange 2 reads landed", "phase 2 butterfly + twiddle + stores issued"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prns", type=int, default=4)
    a = ap.parse_args()
    from bds_amd import native
    s, x, sats, label = bench.build_workload("b1c")
    ctx = native.Context(0)
    ctx.acq_load(s, x)
    ctx.acq_prepare(s)
    prns = list(range(1, a.prns + 1))
    L = native.lib()
    L.bds_debug_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 128)()
    ctx.acq_run(s, prn_list=prns)
    L.bds_debug_phases(buf, 128)  # clear after the warm-up
    ctx.acq_run(s, prn_list=prns)
    t = ctx.timing()
    L.bds_debug_phases(buf, 128)
    v = np.array(list(buf), dtype=np.float64)
    nw = v[24]
    print(f"pair {t['cell_pair_ms']:.3f} ms (rows {t['rows_ms']:.3f}, cols {t['cols_ms']:.3f}) with the phase stamps in")
    print(f"column pass: {int(nw)} waves (one tile each), cycles per wave:")
    tot = 0
    for i, name in enumerate(COLS):
        if "unused" in name:
            continue
        print(f"  {v[i] / nw:9.0f}  {name}")
        tot += v[i] / nw
    print(f"  {tot:9.0f}  total")
    nwr = v[32 + 17]
    pro = v[32 + 16] / nwr
    print(f"row pass: {int(nwr)} waves; prologue {pro:.0f} cycles per wave; per transform (component 0 / component 1), cycles:")
    # transforms per wave: total cells x rows / (waves / 4 waves per workgroup) ... derive from the sums instead: count via cells
    D = t['n_bins']
    nwg_total = 768 * ((D + 33) // 34) * a.prns  # row workgroups of the run (one row, <= 34 cells each)
    ntr = a.prns * D * 768 / nwg_total  # transforms of each component a (sampled) wave takes part in
    tot0 = tot1 = 0
    for i, name in enumerate(ROWS):
        c0, c1 = v[32 + i] / nwr / ntr, v[32 + 8 + i] / nwr / ntr
        print(f"  {c0:8.0f} {c1:8.0f}  {name}")
        tot0 += c0
        tot1 += c1
    print(f"  {tot0:8.0f} {tot1:8.0f}  total")


if __name__ == "__main__":
    main()
