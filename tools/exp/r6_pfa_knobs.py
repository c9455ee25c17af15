#!/usr/bin/env python3
"""Launch knobs of the N-point pair at cfg3 (hooks library): whole bds_acq_run calls, alternating, one box.
    python tools/exp/r6_pfa_knobs.py > profiles/r06_pfa53_knobs.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("BDS_LIB_PATH", os.path.join(ROOT, "bds-3-b1c-b2a-sdr-receiver_amd", "libbds_mi355x_hooks.so"))
import bds_amd  # noqa: E402
import bench  # noqa: E402

s, x, sats, _ = bench.build_workload("b1c")
VARIANTS = [{}, {"BDS_ACQ_PFA_QCHUNK": "2"}, {"BDS_ACQ_PFA_QCHUNK": "4"}, {"BDS_ACQ_PFA_QCHUNK": "8"}, {"BDS_ACQ_PFA_CGRID": "2048"}, {"BDS_ACQ_PFA_CGRID": "4096"},
            {"BDS_ACQ_PFA_CGRID": "16384"}, {"BDS_ACQ_PFA_CGRID": "32768"}, {"BDS_ACQ_LIST_GC": "201"}, {"BDS_ACQ_PAIR_GB": "0"}, {"BDS_ACQ_PAIR_GB": "20"}, {"BDS_ACQ_PAIR_GB": "80"},
            {"BDS_ACQ_PAIR_GB": "auto"}, {"BDS_ACQ_PFA": "0"}, {}]
for env in VARIANTS:
    for k, v in env.items():
        os.environ[k] = v
    c = bds_amd.native.Context(0)
    c.acq_load(s, x)
    c.acq_prepare(s)
    best = None
    for _ in range(4):
        c.acq_run(s)
        t = c.timing()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    c.close()
    for k in env:
        del os.environ[k]
    print(f"{str(env):45s} call {best['total_ms']:.2f} ms  forward {best['forward_ms']:.2f}  search {best['search_ms']:.2f}  refine {best['refine_ms']:.2f}  "
          f"pair {best['cell_pair_ms']:.3f} ms / {best['cells_per_pair']:.0f} cells = {best['cell_pair_ms'] / best['cells_per_pair'] * 201:.3f} ms per 201 (rows {best['rows_ms'] / best['cells_per_pair'] * 201:.3f} + columns {best['cols_ms'] / best['cells_per_pair'] * 201:.3f})  list {best['n_extra']}", flush=True)
