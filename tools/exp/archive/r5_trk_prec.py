#!/usr/bin/env python3
"""Round-5 experiment: closed-loop tracking against the f64 oracle fixtures over the full horizon, per
numerics mode of the run-based correlator (BDS_TRK_PREC: 0 fp32 carrier + fp32 prefix sums, 1 f64 prefix sums, 2 f64 carrier
too, 3 the reference's own trigarg per sample with the library's division and sincos, 4 the same values from the hand-written
division and sin / cos of csrc/bds_strict_math.h -- the default) and per segment length: first epoch at which SURVEY section 8d's closed-loop
tolerances break (I/Q 1e-4 |P|, carrFreq 1e-3 Hz, codeFreq 1e-6 Hz), worst errors over the horizon, and us per epoch on
12 channels at 99.375 MS/s.
    python tools/exp/r5_trk_prec.py [--quick] [prec ...]     -> one JSON line per (fixture, prec, seg) and per (mode, prec, seg) timing"""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("BDS_LIB_PATH", os.path.join(ROOT, "bds-3-b1c-b2a-sdr-receiver_amd", "libbds_mi355x_hooks.so"))  # tuning switches: test-hooks build only
import bds_amd  # noqa: E402

quick = "--quick" in sys.argv
PRECS = tuple(int(a) for a in sys.argv[1:] if a.isdigit()) or (0, 1, 2, 3, 4)  # numerics modes to run
ctx = bds_amd.get_context(0)


def load(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    s = bds_amd.Settings(**json.loads(str(z["settings"])))
    chans = [SimpleNamespace(**c) for c in json.loads(str(z["channels"]))]
    n_epochs, shift = int(z["n_epochs"]), int(z["shift"])
    spc = int(np.floor(s.samplingFreq / (s.codeFreqBasis / s.codeLength) + 0.5))
    n = (n_epochs + 4) * spc + shift
    b = np.roll(z["block"], shift)
    return z, s, chans, np.tile(b, -(-n // b.size))[:n], n_epochs


def first_bad(err, tol):
    bad = np.nonzero(err > tol)[0]
    return int(bad[0]) if bad.size else -1


def set_mode(prec, seg):
    os.environ["BDS_TRK_PREC"] = str(prec)
    if seg:
        os.environ["BDS_TRK_SEG"] = str(seg)
    else:
        os.environ.pop("BDS_TRK_SEG", None)
    ctx.reload_tuning()


fixtures = {n: load(n) for n in ("trk_wb_long", "trk_b2a_long")}
for name, (z, s, chans, x, n_epochs) in fixtures.items():
    mode = str(z["mode"])
    for prec in PRECS:
        for seg in ((0,) if quick else (0, 8 if mode != "B2A" else 16)):
            set_mode(prec, seg)
            got, _ = bds_amd.tracking(x, chans, s, mode=mode)
            dev_ms = ctx.timing()["total_ms"]
            rec = {"fixture": name, "prec": prec, "seg": seg, "us_per_epoch": 1e3 * dev_ms / n_epochs, "channels": []}
            for c, g in enumerate(got):
                p = float(np.hypot(z["I_P"][c].astype(np.float64), z["Q_P"][c].astype(np.float64)).max())
                e_iq = np.zeros(n_epochs)
                for f in ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_P", "Pilot_Q_P"):
                    e_iq = np.maximum(e_iq, np.abs(getattr(g, f) - z[f][c].astype(np.float64)) / p)
                e_cf = np.abs(g.carrFreq - z["carrFreq"][c])
                e_kf = np.abs(g.codeFreq - z["codeFreq"][c])
                e_rc = np.abs(g.remCodePhase - z["remCodePhase"][c])
                rec["channels"].append({
                    "completed": int(g.completed), "absSample_exact": bool(np.array_equal(g.absoluteSample, z["absoluteSample"][c])),
                    "first_bad_iq_1e-4": first_bad(e_iq, 1e-4), "first_bad_carr_1e-3": first_bad(e_cf, 1e-3),
                    "first_bad_code_1e-6": first_bad(e_kf, 1e-6),
                    "max_iq": float(e_iq.max()), "max_carrFreq": float(e_cf.max()), "max_codeFreq": float(e_kf.max()),
                    "max_remCode": float(e_rc.max()), "remCode_at_100": float(e_rc[:100].max()), "remCode_median": float(np.median(e_rc))})
            print(json.dumps(rec), flush=True)

# timing at the full rate: 12 channels, 99.375 MS/s, noise record
for mode in ("WB", "NB", "B2A"):
    epochs = 100 if mode != "B2A" else 400
    if mode == "B2A":
        s = bds_amd.init_settings_b2a(msToProcess=epochs, numberOfChannels=12)
        spc = 99375
    else:
        s = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, msToProcess=epochs * 10, numberOfChannels=12,
                                      pilotTRKflag=2 if mode == "WB" else 1)
        spc = 993750
    rng = np.random.default_rng(1)
    base = 52 * spc
    n = (epochs + 2) * spc
    x = np.clip(np.rint(rng.normal(0, 20, base)), -127, 127).astype(np.int8)
    x = np.tile(x, n // base + 1)[:n]
    ch = [SimpleNamespace(PRN=p, acquiredFreq=s.IF + 100.0 * i, codePhase=float(1000 * i + 1), codeFreq=s.codeFreqBasis, status="T")
          for i, p in enumerate(range(1, 13))]
    for prec in PRECS:
        for seg in (0, 8 if mode != "B2A" else 16):
            set_mode(prec, seg)
            bds_amd.tracking(x, ch, s, mode=mode)
            best = 1e9
            for _ in range(3):
                bds_amd.tracking(x, ch, s, mode=mode)
                best = min(best, ctx.timing()["total_ms"])
            print(json.dumps({"timing": mode, "prec": prec, "seg": seg, "us_per_epoch_12ch_99MSps": 1e3 * best / epochs}), flush=True)
