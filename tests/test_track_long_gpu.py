"""Closed-loop tracking over the reference's full horizon against the oracle (tests/golden/make_long_tracking.py):
3 600 ten-millisecond epochs of B1C wide-band tracking (36 000 ms = BASELINE.json configs[3], 2 channels at 12.5 MS/s)
and 49 000 one-millisecond epochs of B2a tracking, on records where the loops lock.

SURVEY.md section 8d's closed-loop tolerances (I/Q 1e-4 of |P|, carrFreq 1e-3 Hz, codeFreq 1e-6 Hz) are asserted over
ALL 3 600 / 49 000 epochs, with absoluteSample exact.  What makes that possible (round 5): the replica code index of every
sample is ceil(code phase), a discontinuous function of the loop state, so two implementations whose remCodePhase differs
by d disagree on ceil() for some sample after about 1 / (3 N d) epochs, and from then on differ by "flip noise" of ~1e-3 |P|.
The reference's carrier argument trigarg(k) = (carrFreq*2*pi) .* (k ./ fs) + remCarrPhase carries a rounding noise of
~1e-10 rad per sample (one ulp at 1e6 rad); a correlator that forms the carrier any other way -- fp32 (d ~ 1e-9: first flip at
epoch 142 / 1 588) or even an accurate f64 phasor recurrence (d ~ 1e-11: epoch 3 191 on one channel) -- differs from the
reference by that noise.  The default correlator (TrkParams::prec 4) reproduces trigarg(k) bit for bit (k ./ fs by an exactly
rounded reciprocal division, csrc/bds_strict_math.h), takes a library-free f64 sin / cos of it per sample and keeps the prefix
sums in f64: correlator sums 1e-13 of |P| from the oracle, d a few ulps of the code length (~1e-12 chip) or exactly 0, no flip on
either fixture (profiles/r05_trk_prec_first.txt, r05_trk_prec_strict.txt, tools/exp/r5_trk_prec.py).  prec 5 -- one sin / cos per
lane and 16 samples, the other 15 by angle addition with a first-order correction, 12 % faster -- gives the same on these fixtures
and was the default until the whole of cfg4 at full rate was run against the oracle (tests/test_cfg4_gpu.py, profiles/
r05_cfg4_full_vs_c_oracle.txt: 43 200 epoch-channels of 1e6 samples; prec 4 is 1e-13 from the oracle and loses ONE channel to a flip,
prec 5 is 4e-10 and loses six).  The fp32 form (BDS_TRK_PREC=0, 1.5x faster in wide-band mode)
is kept as an option and tested against the bounded flip-noise floor below."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

import bds_amd

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FAST_TIGHT = {"WB": 100, "B2A": 800}  # fp32 carrier (prec 0): epochs before its first ceil() flip (142 / 1 588)


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    s = bds_amd.Settings(**json.loads(str(z["settings"])))
    chans = [SimpleNamespace(**c) for c in json.loads(str(z["channels"]))]
    n_epochs, shift = int(z["n_epochs"]), int(z["shift"])
    spc = int(np.floor(s.samplingFreq / (s.codeFreqBasis / s.codeLength) + 0.5))
    n = (n_epochs + 4) * spc + shift
    b = np.roll(z["block"], shift)
    x = np.tile(b, -(-n // b.size))[:n]  # the record = the committed block repeated
    return z, s, chans, x, n_epochs


def _check(z, got, n_epochs, mode, k, floor):
    """section 8d over the first k epochs; `floor` (x of the tolerances) over the rest"""
    for c, g in enumerate(got):
        assert g.status == "T" and g.completed == n_epochs
        np.testing.assert_array_equal(g.absoluteSample, z["absoluteSample"][c])
        p = float(np.hypot(z["I_P"][c].astype(np.float64), z["Q_P"][c].astype(np.float64)).max())
        for f in ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_P", "Pilot_Q_P"):
            want = z[f][c].astype(np.float64)  # stored as float32: 6e-8 of their magnitude, far inside the tolerances
            np.testing.assert_allclose(getattr(g, f)[:k], want[:k], rtol=0, atol=1e-4 * p, err_msg=f)
            if k < n_epochs:
                np.testing.assert_allclose(getattr(g, f), want, rtol=0, atol=1e-2 * p, err_msg=f)
        # (remCarrPhase: 2 pi x carrFreq tolerance x 10 ms = 6e-5 rad)
        for f, tight, fl in (("carrFreq", 1e-3, 0.06), ("codeFreq", 1e-6, 0.02), ("remCodePhase", 1e-7, 5e-4), ("remCarrPhase", 1e-4, 5e-3)):
            np.testing.assert_allclose(getattr(g, f)[:k], z[f][c][:k], rtol=0, atol=tight, err_msg=f)
            if k < n_epochs:
                np.testing.assert_allclose(getattr(g, f), z[f][c], rtol=0, atol=fl, err_msg=f)
        sig = "B2a_CNo" if mode == "B2A" else "B1C_CNo"
        np.testing.assert_allclose(getattr(g, sig), z["SigCNo"][c], rtol=0, atol=0.05)
        if floor:  # the difference does not grow: the last quarter is no worse than the second one
            d = np.abs(g.I_P - z["I_P"][c].astype(np.float64))
            q = n_epochs // 4
            assert d[3 * q:].max() <= 2.0 * d[q:2 * q].max() + 1e-4 * p
        # the loops are locked over the whole run (the fixture generator asserts the same of the oracle)
        assert np.abs(g.I_P[n_epochs // 2:]).mean() > 3 * np.abs(g.Q_P[n_epochs // 2:]).mean()


@pytest.mark.parametrize("name", ["trk_wb_long", "trk_b2a_long"])
def test_full_horizon_closed_loop(ctx, name):
    """the default (strict) correlator: section 8d over every epoch of the horizon"""
    z, s, chans, x, n_epochs = _load(name)
    mode = str(z["mode"])
    assert n_epochs == {"WB": 3600, "B2A": 49000}[mode]
    got, _ = bds_amd.tracking(x, chans, s, mode=mode)
    _check(z, got, n_epochs, mode, n_epochs, floor=False)


@pytest.mark.parametrize("name", ["trk_wb_long", "trk_b2a_long"])
def test_full_horizon_fast_correlator(ctx, name, monkeypatch):
    """BDS_TRK_PREC=0 (fp32 carrier and prefix sums): section 8d before its first flip, a floor that does not grow after it"""
    z, s, chans, x, n_epochs = _load(name)
    mode = str(z["mode"])
    monkeypatch.setenv("BDS_TRK_PREC", "0")
    ctx.reload_tuning()
    try:
        got, _ = bds_amd.tracking(x, chans, s, mode=mode)
    finally:
        monkeypatch.delenv("BDS_TRK_PREC")
        ctx.reload_tuning()
    _check(z, got, n_epochs, mode, FAST_TIGHT[mode], floor=True)


@pytest.mark.skipif(bool(os.environ.get("BDS_TEST_SKIP_WHOLE_HORIZON")), reason="BDS_TEST_SKIP_WHOLE_HORIZON set (B2a tracking at the reference's own defaults, "
                    "12 channels x 49 000 ms at 99.375 MS/s, against the oracle: a 4.9 GB record and 1-2 min of host time)")
def test_b2a_tracking_at_the_references_defaults_whole_horizon(ctx):
    """B2a/initSettings.m as checked in: fs = 99.375 MS/s, msToProcess = 49 000, 12 channels -- every 1-ms epoch of every channel against
    the float64 oracle (sample loops in C, oracle/c/trk_oracle.c).  Same assertions as tests/test_cfg4_gpu.py's whole-horizon test:
    absoluteSample exact everywhere, SURVEY 8d up to a channel's first ceil() flip, a bounded floor after it."""
    import bench
    from oracle import cfast

    cfast.build()
    n_ep = int(os.environ.get("BDS_TEST_B2A_TRK_EPOCHS", "49000"))
    base = bds_amd.init_settings_b2a()
    assert base.samplingFreq == 99.375e6 and base.msToProcess == 49000 and base.numberOfChannels == 12
    s, x, ch, mode, _, _, _ = bench.track_record("b2a", base, epochs=n_ep)
    got, _ = bds_amd.tracking(x, ch, s, mode=mode)
    ref = cfast.tracking_parallel(x, ch, s, mode=mode)
    fields = ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_P", "Pilot_Q_P")
    n_sep = n_bad = 0
    quiet = [0.0, 0.0, 0.0]
    for c, (r, g) in enumerate(zip(ref, got)):
        assert g.status == "T" and r.status == "T"
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)
        p = np.hypot(r.I_P, r.Q_P).max()
        d_iq = np.max(np.stack([np.abs(getattr(g, f) - getattr(r, f)) for f in fields]), axis=0) / p
        d_carr, d_code = np.abs(g.carrFreq - r.carrFreq), np.abs(g.codeFreq - r.codeFreq)
        bad = (d_iq > 1e-4) | (d_carr > 1e-3) | (d_code > 1e-6)
        if bad.any():
            a = int(np.argmax(bad))
            n_sep += 1
            n_bad += n_ep - a
            print(f"  channel {c} (PRN {r.PRN}): leaves 8d at epoch {a + 1}: first discrepancy {d_iq[a] * p:.1f} (= {d_iq[a]:.2e} of |P|), afterwards worst I/Q "
                  f"{d_iq[a:].max():.2e} of |P|, carrFreq {d_carr[a:].max():.2e} Hz, codeFreq {d_code[a:].max():.2e} Hz")
            assert d_iq[a] * p <= 4 * 127 * 2
            assert d_iq[a:].max() <= 2e-2 and d_carr[a:].max() <= 0.5 and d_code[a:].max() <= 0.05
            ok = slice(0, a)
        else:
            ok = slice(0, n_ep)
        if d_iq[ok].size:
            quiet = [max(quiet[0], float(d_iq[ok].max())), max(quiet[1], float(d_carr[ok].max())), max(quiet[2], float(d_code[ok].max()))]
    total = n_ep * len(ref)
    print(f"B2a tracking at the reference's defaults, 12 channels x {n_ep} epochs x 99.375 MS/s vs the oracle: absoluteSample exact on all {total} "
          f"epoch-channels; {total - n_bad} inside SURVEY 8d (worst there: I/Q {quiet[0]:.2e} of |P|, carrFreq {quiet[1]:.2e} Hz, codeFreq {quiet[2]:.2e} Hz); "
          f"{n_sep} channel(s) separate after a ceil() flip")
    assert n_sep <= 6
