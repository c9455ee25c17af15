#!/usr/bin/env python3
"""Generate the committed golden vectors (inputs + expected outputs) from the float64 oracle.

    python tests/golden/make_golden.py

The reference is MATLAB and cannot run here (no MATLAB/Octave), so these
vectors come from the repo's own restatement (oracle/): they pin the oracle against
regressions and give the GPU tests fixed inputs/outputs that do not depend on the oracle
being importable.  PARITY UNPINNED by the reference itself -- see oracle/__init__.py.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bds_amd  # noqa: E402
from bds_amd import synth  # noqa: E402
from oracle import acquisition as oacq, codes, tracking as otrk  # noqa: E402
from helpers import spc_of, track_case  # noqa: E402


def code_digests():
    s2 = bds_amd.init_settings_b2a()
    s1 = bds_amd.init_settings_b1c()
    out = {}
    for prn in range(1, 64):
        for name, c in (("b2a_data", codes.generate_b2a_data_code(prn, s2)),
                        ("b2a_pilot", codes.generate_b2a_pilot_code(prn, s2)),
                        ("b1c_data", codes.b1c_primary(prn, "data")),
                        ("b1c_pilot", codes.b1c_primary(prn, "pilot"))):
            out[f"{name}_{prn}"] = [codes.octal_digest(c[:24]), codes.octal_digest(c[-24:]), int(c.sum())]
    json.dump(out, open(os.path.join(HERE, "code_digests.json"), "w"), indent=0, sort_keys=True)


def acq_case(name, s, sats, n_samples, seed, fn):
    x = synth.make_if(s, sats, n_samples, seed=seed)
    diag = {}
    r = fn(x.astype(np.float64), s, diag)
    prns = [int(p) for p in s.acqSatelliteList]
    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"), x=x, settings=json.dumps(s.__dict__),
        carrFreq=r.carrFreq, codePhase=r.codePhase, peakMetric=r.peakMetric,
        row_max=np.stack([diag[p]["row_max"] for p in prns]), row_arg=np.stack([diag[p]["row_arg"] for p in prns]),
        peak=np.array([diag[p]["peak"] for p in prns]), fbin=np.array([diag[p]["fbin"] for p in prns]))


def track_golden(name, signal, mode, n_epochs):
    s, x, chans = track_case(signal, mode, n_epochs, seed=77)
    trace = []
    res, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode=mode, trace=trace)
    fields = ["absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L",
              "Pilot_I_P", "Pilot_Q_P", "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt",
              "remCodePhase", "remCarrPhase", "DataCNo", "DataPLD", "PilotCNo", "PilotPLD"]
    if mode == "WB":
        fields += ["Pilot_I_E", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_L"]
    out = {f: np.stack([getattr(r, f) for r in res]) for f in fields}
    out["SigCNo"] = np.stack([getattr(r, "B2a_CNo" if mode == "B2A" else "B1C_CNo") for r in res])
    out["raw_sums"] = np.stack([t["sums"] for t in trace]).reshape(len(chans), n_epochs, 18)
    out["blk"] = np.array([t["blk"] for t in trace]).reshape(len(chans), n_epochs)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), x=x, settings=json.dumps(s.__dict__),
                        channels=json.dumps([c.__dict__ for c in chans]), mode=mode, **out)


if __name__ == "__main__":
    code_digests()
    s = bds_amd.init_settings_b2a(samplingFreq=25e6, IF=6.5e6, acqSatelliteList=[5, 9, 14], acqSearchBand=2000, fineNoncoh=5)
    acq_case("acq_b2a_small", s, [synth.Sat(9, -1230.0, 12345.6, 2.0, 48.0)], 8 * spc_of(s), 2, oacq.acquisition_b2a)
    s = bds_amd.init_settings_b1c(samplingFreq=12.5e6, IF=3.5e6, acqSatelliteList=[3, 7], acqSearchBand=300)
    acq_case("acq_b1c_small", s, [synth.Sat(3, 230.0, 40000.3, 1.0, 45.0)], 4 * spc_of(s), 1, oacq.acquisition_b1c)
    track_golden("trk_b2a_small", "B2A", "B2A", 20)
    track_golden("trk_nb_small", "B1C", "NB", 4)
    track_golden("trk_wb_small", "B1C", "WB", 4)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
