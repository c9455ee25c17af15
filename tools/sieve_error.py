#!/usr/bin/env python3
"""Error of the search grid (the sieve) against the float64 oracle, per storage / arithmetic mode: sets and checks the
refinement tolerance kDelta of bds_acq.hip (the sieve is complete while its error stays below kDelta / 2).

    python tools/sieve_error.py            (GPU box; prints one line per mode and case)

For every (PRN, Doppler bin) row the GPU's row maximum is compared with the oracle's; reported: worst and rms
relative error of the row maxima, worst error relative to the PRN's global maximum (the quantity the tolerance is
defined on), and whether the row argmax agrees."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

os.environ.setdefault("BDS_LIB_PATH", os.path.join(ROOT, "bds-3-b1c-b2a-sdr-receiver_amd", "libbds_mi355x_hooks.so"))  # tuning switches: test-hooks build only
import bds_amd  # noqa: E402
from helpers import medium_b2a, small_b1c  # noqa: E402
from oracle import acquisition as oacq  # noqa: E402

KDELTA = {0: 2e-5, 1: 2e-3}
MODES = (("fp32 arithmetic, fp16 storage (default)", {}), ("fp32 arithmetic, fp32 storage", {"BDS_ACQ_FP16": "0"}),
         ("default storage, wave-private column pass forced", {"BDS_ACQ_WCOLS": "1"}))
for name, fn, ofn in (("B2a 4 PRNs x 26 bins (256 x 1280)", medium_b2a, oacq.acquisition_b2a),
                      ("B1C 3 PRNs x 21 bins (256 x 2048)", small_b1c, oacq.acquisition_b1c)):
    s, x, _ = fn()
    diag = {}
    ofn(x.astype(np.float64), s, diag)
    prns = [int(p) for p in s.acqSatelliteList]
    nb = len(oacq.freq_bins(s))
    for label, env in MODES:
        os.environ.update(env)
        c = bds_amd.native.Context(0)
        c.acq_load(s, x)
        c.acq_prepare(s)
        c.acq_run(s)
        mode = c.timing()["half_storage"]
        rm, ra = c.acq_grid(len(prns), nb)
        c.close()
        for k in env:
            del os.environ[k]
        ref = np.stack([diag[p]["row_max"] for p in prns])
        rel = rm / ref - 1
        glob = np.abs(rm - ref) / ref.max(axis=1, keepdims=True)
        arg = np.mean([ra[i] == diag[p]["row_arg"] for i, p in enumerate(prns)])
        print(f"{name:36s} {label:42s} mode {mode}: row-max rel err max {np.abs(rel).max():.2e} rms {np.sqrt((rel**2).mean()):.2e}; "
              f"vs PRN maximum max {glob.max():.2e}  (kDelta/2 = {KDELTA[mode] / 2:.0e}); row argmax agrees {arg:.2f}")
