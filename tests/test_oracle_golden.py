"""The float64 oracle against the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py) and against the synthetic-IF ground truth (injected delay / Doppler)."""
import json
import os

import numpy as np
import pytest

import bds_amd
from bds_amd import synth
from oracle import acquisition as oacq, tracking as otrk
from types import SimpleNamespace

from helpers import spc_of

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    s = bds_amd.Settings(**json.loads(str(z["settings"])))
    return z, s


@pytest.mark.parametrize("name,fn", [("acq_b2a_small", oacq.acquisition_b2a), ("acq_b1c_small", oacq.acquisition_b1c)])
def test_acquisition_golden(name, fn):
    z, s = load(name)
    diag = {}
    r = fn(z["x"].astype(np.float64), s, diag)
    np.testing.assert_array_equal(r.codePhase, z["codePhase"])
    np.testing.assert_array_equal(r.carrFreq, z["carrFreq"])
    np.testing.assert_allclose(r.peakMetric, z["peakMetric"], rtol=1e-12)
    prns = [int(p) for p in s.acqSatelliteList]
    np.testing.assert_allclose(np.stack([diag[p]["row_max"] for p in prns]), z["row_max"], rtol=1e-10)


def test_acquisition_recovers_injected_satellite():
    s = bds_amd.init_settings_b2a(samplingFreq=25e6, IF=6.5e6, acqSatelliteList=[9, 14], acqSearchBand=2000, fineNoncoh=5)
    sat = synth.Sat(9, -1230.0, 12345.6, 2.0, 48.0)
    x = synth.make_if(s, [sat], 8 * spc_of(s), seed=2)
    r = oacq.acquisition_b2a(x.astype(np.float64), s)
    spc = spc_of(s)
    assert r.carrFreq[13] == 0 and r.carrFreq[8] != 0
    assert abs(r.carrFreq[8] - (s.IF + sat.doppler)) <= 25
    assert abs(((r.codePhase[8] - 1) - sat.delay) % spc) <= 2 or abs(((r.codePhase[8] - 1) - sat.delay) % spc - spc) <= 2
    assert r.peakMetric[8] > s.acqThreshold > r.peakMetric[13]


def test_b1c_acquisition_recovers_injected_satellite():
    s = bds_amd.init_settings_b1c(samplingFreq=12.5e6, IF=3.5e6, acqSatelliteList=[3, 7], acqSearchBand=300)
    sat = synth.Sat(3, 230.0, 40000.3, 1.0, 45.0)
    x = synth.make_if(s, [sat], 4 * spc_of(s), seed=1)
    r = oacq.acquisition_b1c(x.astype(np.float64), s)
    spc = spc_of(s)
    assert r.carrFreq[6] == 0 and abs(r.carrFreq[2] - (s.IF + sat.doppler)) <= 25
    d = ((r.codePhase[2] - 1) - sat.delay) % spc
    assert min(d, spc - d) <= 2


@pytest.mark.parametrize("name", ["trk_b2a_small", "trk_nb_small", "trk_wb_small"])
def test_tracking_golden(name):
    z, s = load(name)
    chans = [SimpleNamespace(**c) for c in json.loads(str(z["channels"]))]
    mode = str(z["mode"])
    trace = []
    res, _ = otrk.tracking(otrk.RawFile(z["x"]), chans, s, mode=mode, trace=trace)
    for f in ("absoluteSample", "codeFreq", "carrFreq", "I_P", "Q_P", "I_E", "Q_L", "Pilot_I_P", "Pilot_Q_P",
              "dllDiscr", "pllDiscr", "remCodePhase", "remCarrPhase"):
        np.testing.assert_allclose(np.stack([getattr(r, f) for r in res]), z[f], rtol=1e-9, atol=1e-9, err_msg=f)
    raw = np.stack([t["sums"] for t in trace]).reshape(z["raw_sums"].shape)
    np.testing.assert_allclose(raw, z["raw_sums"], rtol=1e-9, atol=1e-6)


def test_tracking_conventions_lock():
    """B2a data power lands on I_P, its pilot on Pilot_Q_P; B1C data on I_P, NB pilot
    on Pilot_Q_P, WB composite pilot on Pilot_I_P."""
    from helpers import track_case

    for signal, mode, n in (("B1C", "NB", 12), ("B1C", "WB", 12)):
        s, x, chans = track_case(signal, mode, n)
        res, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode=mode)
        r = res[0]
        tail = slice(n // 2, None)
        assert np.mean(np.abs(r.I_P[tail])) > 3 * np.mean(np.abs(r.Q_P[tail]))
        if mode == "NB":
            assert np.mean(np.abs(r.Pilot_Q_P[tail])) > 3 * np.mean(np.abs(r.Pilot_I_P[tail]))
        else:
            assert np.mean(np.abs(r.Pilot_I_P[tail])) > 3 * np.mean(np.abs(r.Pilot_Q_P[tail]))
        assert r.status == "T" and np.all(np.diff(r.absoluteSample) > 0)


def test_short_file_semantics():
    """B2a/tracking.m:250-254: partial results, later channels untouched, status '-'."""
    from helpers import track_case

    s, x, chans = track_case("B2A", "B2A", 20)
    res, _ = otrk.tracking(otrk.RawFile(x[: 10 * 25000]), chans, s, mode="B2A")
    assert [r.status for r in res] == ["-", "-", "-"]
    assert 0 < np.sum(np.isfinite(res[0].carrFreq)) < 20
    assert not np.any(res[1].I_P) and res[1].PRN is None


def test_loop_coefficients_and_weight():
    s1 = bds_amd.init_settings_b1c()
    s2 = bds_amd.init_settings_b2a()
    # SURVEY.md section 8a row a18
    np.testing.assert_allclose(otrk.calc_loop_coef(1, 0.7, 1.0), (0.279388, 0.74), rtol=1e-5)
    np.testing.assert_allclose(otrk.calc_loop_coef_carr(s1), (0.298598, 4.1472, 28.8), rtol=1e-5)
    np.testing.assert_allclose(otrk.calc_loop_coef(2, 0.7, 1.0), (0.0698469, 0.37), rtol=1e-5)
    np.testing.assert_allclose(otrk.calc_loop_coef_carr(s2), (0.013824, 1.152, 48), rtol=1e-5)
    assert abs(otrk.calc_weighing_factor(s1) - 0.1635) < 5e-4


def test_pre_run_orders_by_peak_metric():
    s = bds_amd.init_settings_b1c(numberOfChannels=3, IF=14.58e6)
    acq = SimpleNamespace(carrFreq=np.array([0, 14.58e6 + 100, 0, 14.58e6 - 250, 14.58e6]),
                          codePhase=np.array([0, 11.0, 0, 22.0, 33.0]), peakMetric=np.array([3.0, 9.0, 2.0, 20.0, 8.0]))
    ch = otrk.pre_run(acq, s)
    assert [c.PRN for c in ch] == [4, 2, 5]
    assert ch[0].codeFreq == s.codeFreqBasis - (-250.0) / s.carrFreqBasis * s.codeFreqBasis
    s2 = bds_amd.init_settings_b2a(numberOfChannels=4)
    ch = otrk.pre_run(acq, s2)
    assert [c.PRN for c in ch] == [4, 2, 5, 0] and ch[0].codeFreq == s2.codeFreqBasis and ch[3].status == "-"
