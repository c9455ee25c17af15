#!/usr/bin/env python3
"""Soak test of the device refinement chain: the small parity cases and cfg2 again and again on fresh contexts; any run that
handed over to the host path (bds_timing.refine_path == 0) is reported with the library's BDS_VERBOSE reason on stderr."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("BDS_LIB_PATH", os.path.join(ROOT, "bds-3-b1c-b2a-sdr-receiver_amd", "libbds_mi355x_hooks.so"))
os.environ["BDS_VERBOSE"] = "1"
import numpy as np  # noqa: E402

import bds_amd  # noqa: E402
import bench  # noqa: E402
from helpers import cfg1_b2a_iq, medium_b2a, small_b1c  # noqa: E402

cases = {"b2a": medium_b2a()[:2] + (False,), "b2a_iq": cfg1_b2a_iq()[:2] + (True,), "cfg2": bench.build_workload("b2a")[:2] + (False,)}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for name, (s, x, cplx) in cases.items():
    first = None
    for it in range(n):
        c = bds_amd.native.Context(0)
        c.acq_load(s, x, is_complex=cplx)
        c.acq_prepare(s)
        for rep in range(2):
            res = c.acq_run(s)
            tm = c.timing()
            if tm["refine_path"] != 1:
                bad += 1
                print(f"{name} iteration {it} run {rep}: refine_path {tm['refine_path']}", flush=True)
            key = np.stack(res[:3]).tobytes()
            if first is None:
                first = key
            elif key != first:
                bad += 1
                print(f"{name} iteration {it} run {rep}: results differ from the first run", flush=True)
        c.close()
    print(name, "done", flush=True)
print("hand-overs / differences:", bad)
