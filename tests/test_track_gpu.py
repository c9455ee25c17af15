"""GPU parity: HIP tracking (through the C ABI) vs the float64 oracle on the same int8 record.

Tolerances: open loop (oracle NCO state injected per epoch) correlator
sums <= 1e-6 of |P|; closed loop I/Q <= 1e-4 of |P|, carrFreq <= 1e-3 Hz, codeFreq <= 1e-6 Hz,
absoluteSample exact.
"""
import numpy as np
import pytest

import bds_amd
from oracle import tracking as otrk

from helpers import track_case

pytestmark = pytest.mark.gpu

CASES = [("B2A", "B2A", 60, False), ("B1C", "NB", 12, False), ("B1C", "WB", 12, False),
         # fileType 2: interleaved I/Q record (tracking.m:132-136,242-246)
         ("B2A", "B2A", 40, True), ("B1C", "NB", 8, True), ("B1C", "WB", 8, True)]


@pytest.mark.parametrize("signal,mode,n_epochs,iq", CASES)
def test_open_loop_correlators(ctx, signal, mode, n_epochs, iq):
    s, x, chans = track_case(signal, mode, n_epochs, iq=iq)
    trace = []
    otrk.tracking(otrk.RawFile(x), chans, s, mode=mode, trace=trace)
    assert len(trace) == n_epochs * len(chans)
    for k in (1, n_epochs // 2, n_epochs):
        rows = [t for t in trace if t["k"] == k]
        prn = [chans[t["ch"]].PRN for t in rows]
        st = [[t["pos"], t["blk"], t["rem"], t["codeFreq"], t["remCarr"], t["carrFreq"]] for t in rows]
        got = ctx.track_correlate(s, x, prn, st)
        for g, t in zip(got, rows):
            p = np.hypot(t["sums"][2], t["sums"][3])
            np.testing.assert_allclose(g, t["sums"], rtol=0, atol=1e-6 * p)


@pytest.mark.parametrize("signal,mode,n_epochs,iq", CASES)
def test_closed_loop_tracking(ctx, signal, mode, n_epochs, iq):
    s, x, chans = track_case(signal, mode, n_epochs, iq=iq)
    ref, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode=mode)
    got, _ = bds_amd.tracking(x, chans, s, mode=mode)
    for r, g in zip(ref, got):
        assert g.status == r.status == "T" and g.PRN == r.PRN
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)
        p = np.hypot(r.I_P, r.Q_P).max()
        for f in ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_P", "Pilot_Q_P"):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-4 * p, err_msg=f)
        if mode == "WB":
            for f in ("Pilot_I_E", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_L"):
                np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-4 * p, err_msg=f)
        np.testing.assert_allclose(g.carrFreq, r.carrFreq, rtol=0, atol=1e-3)
        np.testing.assert_allclose(g.codeFreq, r.codeFreq, rtol=0, atol=1e-6)
        np.testing.assert_allclose(g.remCodePhase, r.remCodePhase, rtol=0, atol=1e-7)
        np.testing.assert_allclose(g.remCarrPhase, r.remCarrPhase, rtol=0, atol=1e-6)
        for f, tol in (("dllDiscr", 1e-6), ("dllDiscrFilt", 1e-6), ("pllDiscr", 1e-6), ("pllDiscrFilt", 1e-3)):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=tol, err_msg=f)
        cn = "B2a_CNo" if mode == "B2A" else "B1C_CNo"
        for f in ("DataCNo", "PilotCNo", cn):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-3, err_msg=f)
        for f in ("DataPLD", "PilotPLD"):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-5, err_msg=f)


def test_short_file_returns_partial_results(ctx):
    """B2a/tracking.m:250-254: at end of file the first channel keeps what it has, later channels are
    never started, status stays '-'."""
    s, x, chans = track_case("B2A", "B2A", 40)
    spc = 25000
    x = x[: 25 * spc]
    ref, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode="B2A")
    got, _ = bds_amd.tracking(x, chans, s, mode="B2A")
    assert [g.status for g in got] == [r.status for r in ref] == ["-", "-", "-"]
    done = int(np.sum(np.isfinite(ref[0].carrFreq)))
    assert got[0].completed == done and 0 < done < 40
    np.testing.assert_array_equal(got[0].absoluteSample, ref[0].absoluteSample)
    np.testing.assert_allclose(got[0].I_P, ref[0].I_P, atol=1e-4 * np.abs(ref[0].I_P).max())
    for c in (1, 2):
        assert not np.any(got[c].I_P) and np.all(np.isinf(got[c].carrFreq))


def test_all_channels_idle(ctx):
    """preRun leaves PRN = 0 in channels it could not fill (preRun.m:46-56): tracking skips them
    (tracking.m:141) and returns the untouched result template."""
    from types import SimpleNamespace

    s, x, chans = track_case("B2A", "B2A", 10)
    idle = [SimpleNamespace(PRN=0, acquiredFreq=0.0, codePhase=0.0, codeFreq=0.0, status="-") for _ in chans]
    ref, _ = otrk.tracking(otrk.RawFile(x), idle, s, mode="B2A")
    got, _ = bds_amd.tracking(x, idle, s, mode="B2A")
    for r, g in zip(ref, got):
        assert g.status == r.status == "-" and g.PRN is None and r.PRN is None and g.completed == 0  # PRN is a lazily added field (tracking.m:144)
        np.testing.assert_array_equal(g.I_P, r.I_P)
        np.testing.assert_array_equal(g.carrFreq, r.carrFreq)
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)


@pytest.mark.parametrize("signal,mode", [("B2A", "B2A"), ("B1C", "NB")])
def test_tracking_without_pilot(ctx, signal, mode):
    """pilotTRKflag = 0: data-only discriminators, and the Pilot_* / PilotCNo / *_CNo fields are not created at all
    (B2a/tracking.m:70-73,89-93; NB_tracking.m:78-81,98-102)."""
    s, x, chans = track_case(signal, mode, 16)
    s = s.copy(pilotTRKflag=0)
    ref, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode=mode)
    got, _ = bds_amd.tracking(x, chans, s, mode=mode)
    for r, g in zip(ref, got):
        assert g.status == r.status == "T"
        assert not hasattr(g, "Pilot_I_P") and not hasattr(r, "Pilot_I_P") and not hasattr(g, "PilotCNo")
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)
        p = np.hypot(r.I_P, r.Q_P).max()
        for f in ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L"):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-4 * p, err_msg=f)
        np.testing.assert_allclose(g.carrFreq, r.carrFreq, rtol=0, atol=1e-3)
        np.testing.assert_allclose(g.codeFreq, r.codeFreq, rtol=0, atol=1e-6)
        np.testing.assert_allclose(g.DataCNo, r.DataCNo, rtol=0, atol=1e-3)


def test_only_the_window_of_a_long_record_is_loaded(ctx, tmp_path):
    """The reference streams blksize samples per epoch (tracking.m:237-240); here only the window the channels can
    touch goes to HBM, so a recording much longer than msToProcess costs nothing extra -- and skipNumberOfBytes moves
    the window.  Results equal those on the short record; end of file is still the real end of the file."""
    n_epochs = 20
    s, x, chans = track_case("B2A", "B2A", n_epochs)
    spc = 25000
    rng = np.random.default_rng(9)
    junk = lambda n: np.clip(np.rint(rng.normal(0, 20, n)), -127, 127).astype(np.int8)  # noqa: E731
    skip = 7 * spc
    long_rec = np.concatenate([junk(skip), x, junk(60 * spc)])
    want, _ = bds_amd.tracking(x, chans, s, mode="B2A")
    assert ctx.track_loaded_bytes() <= x.size
    s_long = s.copy(skipNumberOfBytes=skip)
    path = tmp_path / "long_record.bin"
    long_rec.tofile(path)
    for source in (long_rec, str(path)):
        got, _ = bds_amd.tracking(source, chans, s_long, mode="B2A")
        assert ctx.track_loaded_bytes() < 0.45 * long_rec.size  # ~ (n_epochs + delays) code periods of 90
        for g, w in zip(got, want):
            assert g.status == "T"
            np.testing.assert_array_equal(g.absoluteSample, w.absoluteSample + skip)  # ftell-based offsets are absolute
            for f in ("I_P", "Q_P", "carrFreq", "codeFreq", "remCodePhase", "DataCNo"):
                np.testing.assert_array_equal(getattr(g, f), getattr(w, f), err_msg=f)


def test_block_longer_than_the_correlate_grid(ctx, monkeypatch):
    """The correlate grid is sized from the slowest channel's code rate; a block that outgrows it (a diverging DLL, an
    odd channel.codeFreq) is walked by the same workgroups in further strides instead of being mistaken for a short
    read.  Forced here with one workgroup per channel (test hook): results must not change."""
    s, x, chans = track_case("B1C", "WB", 6)
    want, _ = bds_amd.tracking(x, chans, s, mode="WB")
    monkeypatch.setenv("BDS_TRK_NBLOCKS", "2")
    ctx.reload_tuning()
    try:
        got, _ = bds_amd.tracking(x, chans, s, mode="WB")
    finally:
        monkeypatch.delenv("BDS_TRK_NBLOCKS")
        ctx.reload_tuning()
    for g, w in zip(got, want):
        assert g.status == "T"
        np.testing.assert_array_equal(g.absoluteSample, w.absoluteSample)
        p = np.hypot(w.I_P, w.Q_P).max()
        for f in ("I_P", "Q_P", "Pilot_I_P", "Pilot_Q_E"):  # fp32 partial sums over more samples per thread: 1e-6 of |P|
            np.testing.assert_allclose(getattr(g, f), getattr(w, f), rtol=0, atol=2e-6 * p, err_msg=f)
        np.testing.assert_allclose(g.carrFreq, w.carrFreq, rtol=0, atol=1e-4)


@pytest.mark.parametrize("signal,mode,fs,IF", [("B2A", "B2A", 12e6, 3e6), ("B1C", "WB", 7.5e6, 2e6), ("B1C", "WB", 3.1e6, 0.8e6)])
def test_low_sampling_rates_and_both_correlators(ctx, monkeypatch, signal, mode, fs, IF):
    """The run-based correlator stages a pass's slice of the code tables in LDS when it fits (every other case of this
    file); at low sampling rates a pass spans more code units than the slice holds (look-ups stay in global memory) and,
    below the BOC(6,1) rate, a sample step skips units (several index steps share one first sample).  Open-loop sums
    against the oracle, and against the per-sample correlator (BDS_TRK_PERSAMPLE) on the same states."""
    n_epochs = 6 if signal == "B1C" else 20
    s, x, chans = track_case(signal, mode, n_epochs, fs=fs, IF=IF)
    trace = []
    otrk.tracking(otrk.RawFile(x), chans, s, mode=mode, trace=trace)
    rows = [t for t in trace if t["k"] in (1, n_epochs)]
    prn = [chans[t["ch"]].PRN for t in rows]
    st = [[t["pos"], t["blk"], t["rem"], t["codeFreq"], t["remCarr"], t["carrFreq"]] for t in rows]
    got = ctx.track_correlate(s, x, prn, st)
    monkeypatch.setenv("BDS_TRK_PERSAMPLE", "1")
    ctx.reload_tuning()
    try:
        per_sample = ctx.track_correlate(s, x, prn, st)
    finally:
        monkeypatch.delenv("BDS_TRK_PERSAMPLE")
        ctx.reload_tuning()
    for g, q, t in zip(got, per_sample, rows):
        p = np.hypot(t["sums"][2], t["sums"][3])
        np.testing.assert_allclose(g, t["sums"], rtol=0, atol=1e-6 * p)
        np.testing.assert_allclose(g, q, rtol=0, atol=1e-6 * p)


@pytest.mark.parametrize("mode,iq", [("WB", False), ("NB", True), ("B2A", False), ("B2A", True)])
def test_random_states_at_full_rate_both_correlators(ctx, monkeypatch, mode, iq):
    """99.375 MS/s (the rate of BASELINE.json's configs: code-table slices in LDS, ~21 half-chip and ~127 BOC(6,1) steps
    per replica and pass): random open-loop states -- code phase, code / carrier frequency, carrier phase, start sample,
    block lengths that end inside a segment -- through the run-based and the per-sample correlator.  Both evaluate the
    reference's index expression for every sample, so the sums may differ by the summation order only."""
    rng = np.random.default_rng(77)
    if mode == "B2A":
        s = bds_amd.init_settings_b2a(msToProcess=4, numberOfChannels=4, fileType=2 if iq else 1)
        spc, L = 99375, 10230
    else:
        s = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, msToProcess=40, numberOfChannels=4,
                                      pilotTRKflag=2 if mode == "WB" else 1, fileType=2 if iq else 1)
        spc, L = 993750, 10230
    n = 3 * spc
    x = np.clip(np.rint(rng.normal(0, 25, n * (2 if iq else 1))), -127, 127).astype(np.int8)
    prn, st = [], []
    for c in range(8):
        code_freq = s.codeFreqBasis * (1 + rng.uniform(-3e-6, 3e-6))
        rem = rng.uniform(-0.4, 0.4) if c else 0.0
        blk = int(np.ceil((L - rem) / (code_freq / s.samplingFreq)))
        if c % 3 == 2:
            blk -= int(rng.integers(1, 5000))  # an arbitrary block length: the last pass ends inside a segment
        prn.append(int(rng.integers(1, 64)))
        st.append([int(rng.integers(0, spc)), blk, rem, code_freq, rng.uniform(0, 2 * np.pi), s.IF + rng.uniform(-5000, 5000)])
    runs = ctx.track_correlate(s, x, prn, st)
    monkeypatch.setenv("BDS_TRK_PERSAMPLE", "1")
    ctx.reload_tuning()
    try:
        per_sample = ctx.track_correlate(s, x, prn, st)
    finally:
        monkeypatch.delenv("BDS_TRK_PERSAMPLE")
        ctx.reload_tuning()
    scale = np.abs(per_sample).max() + 127.0 * np.sqrt(spc)  # noise-only record: the size of a correlator sum
    np.testing.assert_allclose(runs, per_sample, rtol=0, atol=2e-7 * scale)
    assert np.abs(per_sample[:, :6]).max() > 0
