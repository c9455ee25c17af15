"""The C restatement of the coarse search (oracle/c/acq_oracle.c, test infrastructure) against the NumPy oracle
(oracle/acquisition.py) and against the committed golden vectors: two restatements written separately from the same .m
lines (B1C/acquisition.m:191-232, B2a/acquisition.m:187-221) with different transforms (own mixed-radix Stockham / pocketfft)
and different sin / cos (C library / NumPy) agree to ~1e-12 -- rounding, not algorithm."""
import json
import os

import numpy as np
import pytest

import bds_amd
from oracle import acquisition as oacq
from oracle import cfast

from helpers import as_complex, cfg1_b2a, cfg1_b2a_iq, small_b1c, small_b1c_iq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def _built():
    cfast.build()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 16, 45, 53, 106, 2 * 3 * 5 * 53, 1280, 4096, 198750])
def test_transform_against_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.normal(size=n) + 1j * rng.normal(size=n)
    f = np.fft.fft(x)
    np.testing.assert_allclose(cfast.fft(x), f, rtol=0, atol=4e-15 * np.abs(f).max() * max(1.0, np.log2(n)))
    i = np.fft.ifft(x)
    np.testing.assert_allclose(cfast.fft(x, inverse=True), i, rtol=0, atol=4e-15 * np.abs(i).max() * max(1.0, np.log2(n)))


def _both(x, s, prn):
    gen = oacq.b1c_coarse_rows if str(s.signal).upper() == "B1C" else oacq.b2a_coarse_rows
    ref = np.stack([row for _, row in gen(x, s, prn)])
    rm, ra, cm, rows = cfast.coarse_rows(x, s, prn, want_rows=True, threads=2)
    return ref, rm, ra, cm, rows


@pytest.mark.parametrize("case", ["b2a", "b2a_iq", "b1c", "b1c_iq", "b1c_nopilot"])
def test_rows_against_the_numpy_oracle(case):
    if case == "b2a":
        s, x, _ = cfg1_b2a()
        x = x.astype(np.float64)
    elif case == "b2a_iq":
        s, xi, _ = cfg1_b2a_iq()
        x = as_complex(xi)
    elif case == "b1c":
        s, x, _ = small_b1c()
        x = x.astype(np.float64)
    elif case == "b1c_iq":
        s, xi, _ = small_b1c_iq()
        x = as_complex(xi)
    else:
        s, x, _ = small_b1c()
        s = s.copy(pilotACQflag=0)
        x = x.astype(np.float64)
    prn = int(s.acqSatelliteList[0])
    ref, rm, ra, cm, rows = _both(x, s, prn)
    scale = ref.max()
    np.testing.assert_allclose(rows, ref, rtol=0, atol=1e-11 * scale)
    np.testing.assert_allclose(rm, ref.max(axis=1), rtol=1e-11)
    np.testing.assert_allclose(cm, ref.max(axis=0), rtol=0, atol=1e-11 * scale)
    # first index of the maximum (MATLAB's max): the same lag unless two lags tie within the rounding
    arg = ref.argmax(axis=1)
    same = ra == arg
    assert np.all(same | (np.abs(ref[np.arange(len(arg)), ra] - ref.max(axis=1)) <= 1e-11 * scale))
    assert same.mean() > 0.9


def test_bin_subsets_accumulate_the_column_maximum():
    s, x, _ = small_b1c()
    x = x.astype(np.float64)
    prn = int(s.acqSatelliteList[0])
    d = len(oacq.freq_bins(s))
    rm_all, ra_all, cm_all, _ = cfast.coarse_rows(x, s, prn, threads=1)
    cm = None
    rm = np.empty(d)
    for b0 in range(0, d, 4):
        bins = range(b0, min(d, b0 + 4))
        r, a, cm, _ = cfast.coarse_rows(x, s, prn, bins=bins, col_max=cm, threads=3)
        rm[b0:b0 + len(r)] = r
    np.testing.assert_array_equal(rm, rm_all)      # rows do not depend on the thread count or the subset
    np.testing.assert_array_equal(cm, cm_all)


@pytest.mark.parametrize("name", ["acq_b2a_small", "acq_b1c_small"])
def test_golden_row_maxima(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    s = bds_amd.Settings(**json.loads(str(z["settings"])))
    x = z["x"].astype(np.float64)
    prns = [int(p) for p in s.acqSatelliteList]
    got = np.stack([cfast.coarse_rows(x, s, p, threads=2)[0] for p in prns])
    np.testing.assert_allclose(got, z["row_max"], rtol=1e-10)


@pytest.mark.parametrize("name,fn", [("acq_b2a_small", oacq.acquisition_b2a), ("acq_b1c_small", oacq.acquisition_b1c)])
def test_acq_results_on_the_c_rows_match_the_golden_vectors(name, fn):
    """acqResults with the Doppler rows from the C restatement and the rest from the NumPy oracle = the committed vectors."""
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    s = bds_amd.Settings(**json.loads(str(z["settings"])))
    r = fn(z["x"].astype(np.float64), s, coarse=cfast.backend(threads=2))
    np.testing.assert_array_equal(r.codePhase, z["codePhase"])
    np.testing.assert_array_equal(r.carrFreq, z["carrFreq"])
    np.testing.assert_allclose(r.peakMetric, z["peakMetric"], rtol=1e-11)


@pytest.mark.parametrize("signal,mode,n_epochs,iq", [("B2A", "B2A", 30, False), ("B1C", "NB", 6, False), ("B1C", "WB", 6, False),
                                                      ("B2A", "B2A", 20, True), ("B1C", "WB", 5, True)])
def test_tracking_epochs_in_c_against_the_numpy_oracle(signal, mode, n_epochs, iq):
    """oracle/c/trk_oracle.c (one epoch's sample loop: code indices, carrier, the 18 correlator sums) behind the oracle's loop filters
    against the all-NumPy oracle, closed loop: the same trajectories to rounding (different sin / cos and summation order)."""
    from oracle import tracking as otrk

    from helpers import track_case

    s, x, chans = track_case(signal, mode, n_epochs, iq=iq)
    ref, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode=mode)
    got = cfast.tracking_parallel(x, chans, s, mode=mode, threads=3)
    for r, g in zip(ref, got):
        assert g.status == r.status == "T" and g.PRN == r.PRN
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)
        p = np.hypot(r.I_P, r.Q_P).max()
        fields = ["I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_P", "Pilot_Q_P"]
        if mode == "WB":
            fields += ["Pilot_I_E", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_L"]
        for f in fields:
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-9 * p, err_msg=f)
        np.testing.assert_allclose(g.carrFreq, r.carrFreq, rtol=0, atol=1e-8)
        np.testing.assert_allclose(g.codeFreq, r.codeFreq, rtol=0, atol=1e-9)
        np.testing.assert_allclose(g.remCodePhase, r.remCodePhase, rtol=0, atol=1e-9)
        np.testing.assert_allclose(g.remCarrPhase, r.remCarrPhase, rtol=0, atol=1e-8)
        cn = "B2a_CNo" if mode == "B2A" else "B1C_CNo"
        for f in ("DataCNo", "PilotCNo", cn, "DataPLD", "PilotPLD"):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-7, err_msg=f)
