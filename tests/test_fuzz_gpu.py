"""Seeded random configurations against the float64 oracle (same tolerances as the fixed cases:
codePhase / carrFreq exact, peakMetric 1e-6; tracking I/Q 1e-4 of |P|, absoluteSample exact).
Reduced sampling rates keep the oracle at a few seconds per case."""
from types import SimpleNamespace

import numpy as np
import pytest

import bds_amd
from bds_amd import synth
from oracle import acquisition as oacq, tracking as otrk

from helpers import as_complex, spc_of

pytestmark = pytest.mark.gpu


def _acq_case(seed):
    rng = np.random.default_rng(1000 + seed)
    b1c = bool(seed % 2)
    fs = float(rng.choice([10.0e6, 12.5e6, 16.0e6, 20.46e6, 25.0e6]))
    IF = float(rng.choice([2.5e6, 3.5e6, 4.092e6]))
    iq = bool(rng.integers(0, 2))
    prns = sorted(int(p) for p in rng.choice(np.arange(1, 64), size=4, replace=False))
    kw = dict(samplingFreq=fs, IF=IF, acqSatelliteList=prns, fileType=2 if iq else 1,
              acqSearchBand=float(rng.choice([300, 500, 1000])), acqStep=float(rng.choice([50, 100, 250])))
    if b1c:
        s = bds_amd.init_settings_b1c(acqCohT=int(rng.choice([2, 5, 8, 10])), pilotACQflag=int(rng.integers(0, 2)), **kw)
        n_codes = 3
    else:
        s = bds_amd.init_settings_b2a(fineNoncoh=int(rng.choice([3, 7, 10])), **kw)
        n_codes = int(s.fineNoncoh) + 3
    spc = spc_of(s)
    present = prns[:2]
    sats = [synth.Sat(p, float(rng.uniform(-0.9, 0.9) * s.acqSearchBand), float(rng.uniform(0, spc)),
                      float(rng.uniform(0, 6.28)), float(rng.uniform(43, 50))) for p in present]
    x = synth.make_if(s, sats, n_codes * spc, seed=2000 + seed, iq_sign=-1 if iq else 0)
    return s, (as_complex(x) if iq else x), sats


@pytest.mark.parametrize("seed", range(10))
def test_random_acquisition_configs(ctx, seed):
    s, x, sats = _acq_case(seed)
    fn = oacq.acquisition_b1c if str(s.signal).upper() == "B1C" else oacq.acquisition_b2a
    ref = fn(x.astype(np.complex128 if np.iscomplexobj(x) else np.float64), s)
    got = bds_amd.acquisition(x, s, verbose=False)
    np.testing.assert_array_equal(got.codePhase, ref.codePhase)
    np.testing.assert_array_equal(got.carrFreq, ref.carrFreq)
    np.testing.assert_allclose(got.peakMetric, ref.peakMetric, rtol=1e-6, atol=0)


_WAVE_PLANS = ["512x4096", "768x4096", "1024x4096", "768x2048", "1024x3072", "1024x2048"]  # all hold the largest case (0.75 M points)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("BDS_FUZZ_WAVE_SEEDS", "6"))))
def test_random_configs_on_the_wave_private_plans(ctx, monkeypatch, seed):
    """The same random configurations with the factorisation forced onto the plans that take the wave-private passes
    (column lengths 512 / 768 / 1024: k_cols_wave_f; 4096-point rows: k_rows_wave_f) -- at these sampling rates the planner
    itself would pick the small 256-column plans.  (BDS_FUZZ_WAVE_SEEDS widens the range for a soak run.)"""
    s, x, sats = _acq_case(100 + seed)
    monkeypatch.setenv("BDS_ACQ_FORCE_L1L2", _WAVE_PLANS[seed % len(_WAVE_PLANS)])
    ctx.reload_tuning()
    try:
        fn = oacq.acquisition_b1c if str(s.signal).upper() == "B1C" else oacq.acquisition_b2a
        ref = fn(x.astype(np.complex128 if np.iscomplexobj(x) else np.float64), s)
        got = bds_amd.acquisition(x, s, verbose=False)
        l1, l2 = (int(v) for v in _WAVE_PLANS[seed % len(_WAVE_PLANS)].split("x"))
        assert ctx.timing()["fft_len"] == l1 * l2
        np.testing.assert_array_equal(got.codePhase, ref.codePhase)
        np.testing.assert_array_equal(got.carrFreq, ref.carrFreq)
        np.testing.assert_allclose(got.peakMetric, ref.peakMetric, rtol=1e-6, atol=0)
    finally:
        monkeypatch.delenv("BDS_ACQ_FORCE_L1L2")
        ctx.reload_tuning()


def _trk_case(seed):
    rng = np.random.default_rng(3000 + seed)
    mode = ["B2A", "NB", "WB"][seed % 3]
    iq = bool(rng.integers(0, 2))
    fs = float(rng.choice([10.0e6, 12.5e6, 20.0e6]))
    nch = int(rng.integers(2, 5))
    prns = [int(p) for p in rng.choice(np.arange(1, 64), size=nch, replace=False)]
    if mode == "B2A":
        n_ep = int(rng.integers(25, 45))
        s = bds_amd.init_settings_b2a(samplingFreq=fs, IF=2.5e6, msToProcess=n_ep, numberOfChannels=nch + 1,
                                      CNoInterval=int(rng.choice([5, 10])), fileType=2 if iq else 1,
                                      pilotTRKflag=int(rng.integers(0, 2)))
        signal = "B2A"
    else:
        n_ep = int(rng.integers(5, 9))
        s = bds_amd.init_settings_b1c(samplingFreq=fs, IF=2.5e6, msToProcess=n_ep * 10, numberOfChannels=nch + 1,
                                      pilotTRKflag={"NB": 1, "WB": 2}[mode], CNoInterval=int(rng.choice([2, 4])),
                                      FEBW=float(rng.choice([4e6, 10e6])), fileType=2 if iq else 1)
        signal = "B1C"
    spc = spc_of(s)
    sats = [synth.Sat(p, float(rng.uniform(-3000, 3000)), float(rng.uniform(10, spc - 10)), float(rng.uniform(0, 6.28)),
                      float(rng.uniform(44, 50))) for p in prns]
    x = synth.make_if(s, sats, (n_ep + 3) * spc, seed=4000 + seed, iq_sign=(-1 if signal == "B2A" else 1) if iq else 0)
    chans = []
    for sat in sats:
        cf = s.IF + round(sat.doppler / 25) * 25
        code_freq = (s.codeFreqBasis - (cf - s.IF) / s.carrFreqBasis * s.codeFreqBasis) if signal == "B1C" else s.codeFreqBasis
        chans.append(SimpleNamespace(PRN=sat.prn, acquiredFreq=float(cf), codePhase=float(int(np.ceil(sat.delay)) + 1),
                                     codeFreq=float(code_freq), status="T"))
    chans.append(SimpleNamespace(PRN=0, acquiredFreq=0.0, codePhase=0.0, codeFreq=0.0, status="-"))  # idle channel
    return s, x, chans, mode


@pytest.mark.parametrize("seed", range(9))
def test_random_tracking_configs(ctx, seed):
    s, x, chans, mode = _trk_case(seed)
    ref, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode=mode)
    got, _ = bds_amd.tracking(x, chans, s, mode=mode)
    for r, g in zip(ref, got):
        assert g.status == r.status and g.PRN == r.PRN
        if r.PRN == 0:
            continue
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)
        p = np.hypot(r.I_P, r.Q_P).max()
        for f in ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L"):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-4 * p, err_msg=f)
        for f in ("Pilot_I_P", "Pilot_Q_P", "Pilot_I_E", "Pilot_Q_L"):
            if hasattr(r, f) and hasattr(g, f):
                np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-4 * p, err_msg=f)
        np.testing.assert_allclose(g.carrFreq, r.carrFreq, rtol=0, atol=1e-3)
        np.testing.assert_allclose(g.codeFreq, r.codeFreq, rtol=0, atol=1e-6)
        np.testing.assert_allclose(g.DataCNo, r.DataCNo, rtol=0, atol=1e-3)
