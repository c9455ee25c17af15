"""acquisition -> preRun -> tracking chained on the GPU, against the same chain run through the oracle
(reduced sampling rate so the float64 oracle finishes in seconds)."""
import numpy as np
import pytest

import bds_amd
from bds_amd import synth
from oracle import acquisition as oacq, tracking as otrk

from helpers import spc_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("signal", ["B2A", "B1C"])
def test_acquire_prerun_track_chain(ctx, signal):
    if signal == "B2A":
        s = bds_amd.init_settings_b2a(samplingFreq=25e6, IF=6.5e6, acqSatelliteList=[5, 9, 19, 33], acqSearchBand=2500,
                                      fineNoncoh=5, msToProcess=40, numberOfChannels=3, CNoInterval=20)
        sats = [synth.Sat(9, -1230.0, 12345.6, 2.0, 50.0), synth.Sat(19, 2210.0, 3001.2, 0.4, 47.0)]
        n_codes, acq_codes, mode, oa = 60, 8, "B2A", oacq.acquisition_b2a
    else:
        s = bds_amd.init_settings_b1c(samplingFreq=12.5e6, IF=3.5e6, acqSatelliteList=[3, 7, 12], acqSearchBand=600,
                                      msToProcess=60, numberOfChannels=3, pilotTRKflag=2, CNoInterval=3, FEBW=10e6)
        sats = [synth.Sat(3, 230.0, 40000.3, 1.0, 48.0), synth.Sat(12, -410.0, 99000.8, 2.0, 46.0)]
        n_codes, acq_codes, mode, oa = 10, 4, "WB", oacq.acquisition_b1c
    spc = spc_of(s)
    x = synth.make_if(s, sats, n_codes * spc, seed=123)
    block = x[: acq_codes * spc]
    acq_ref = oa(block.astype(np.float64), s)
    acq_gpu = bds_amd.acquisition(block, s, verbose=False)
    for f in ("carrFreq", "codePhase"):
        np.testing.assert_array_equal(getattr(acq_gpu, f), getattr(acq_ref, f))
    np.testing.assert_allclose(acq_gpu.peakMetric, acq_ref.peakMetric, rtol=1e-6)
    ch_ref = otrk.pre_run(acq_ref, s)
    ch_gpu = bds_amd.pre_run(acq_gpu, s)
    assert [(c.PRN, c.codePhase, c.acquiredFreq, c.codeFreq, c.status) for c in ch_ref] == \
           [(c.PRN, c.codePhase, c.acquiredFreq, c.codeFreq, c.status) for c in ch_gpu]
    assert sorted(c.PRN for c in ch_gpu if c.PRN) == sorted(sat.prn for sat in sats)
    trk_ref, _ = otrk.tracking(otrk.RawFile(x), ch_ref, s, mode=mode)
    trk_gpu, _ = bds_amd.tracking(x, ch_gpu, s, mode=mode)
    for r, g in zip(trk_ref, trk_gpu):
        assert g.status == r.status and g.PRN == r.PRN
        if r.PRN is None:  # unused channel: template values only
            assert not np.any(g.I_P) and np.all(np.isinf(g.carrFreq))
            continue
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)
        p = np.hypot(r.I_P, r.Q_P).max()
        for f in ("I_P", "Q_P", "I_E", "Q_E", "I_L", "Q_L", "Pilot_I_P", "Pilot_Q_P"):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-4 * p, err_msg=f)
        np.testing.assert_allclose(g.carrFreq, r.carrFreq, rtol=0, atol=1e-3)
        np.testing.assert_allclose(g.codeFreq, r.codeFreq, rtol=0, atol=1e-6)
        np.testing.assert_allclose(g.DataCNo, r.DataCNo, rtol=0, atol=1e-3)


def _chan(c):
    return (int(c.PRN), float(c.codePhase), float(c.acquiredFreq), float(c.codeFreq), c.status if isinstance(c.status, str) else chr(c.status))


def test_device_pre_run_equals_the_host_loop(ctx):
    """bds_pre_run_device (one wave, stable rank by counting) against bds_pre_run (preRun.m:61-76 as a host loop): random
    metrics with ties, more and fewer detections than channels, both receivers' codeFreq rules."""
    rng = np.random.default_rng(5)
    for signal, init in (("B1C", bds_amd.init_settings_b1c), ("B2A", bds_amd.init_settings_b2a)):
        for nch, ndet in ((12, 20), (12, 5), (3, 3), (1, 0), (10, 63)):
            s = init(numberOfChannels=nch)
            pm = np.round(rng.uniform(0.5, 9.0, 63), 1)  # one decimal: ties are common
            carr = np.zeros(63)
            cph = np.zeros(63)
            det = rng.choice(63, ndet, replace=False)
            carr[det] = s.IF + rng.integers(-200, 200, ndet) * 25.0
            cph[det] = rng.integers(1, 90000, ndet).astype(float)
            want = bds_amd.native.pre_run(s, carr, cph, pm)
            got = ctx.pre_run_device(s, carr, cph, pm)
            assert [_chan(c) for c in got] == [_chan(c) for c in want], (signal, nch, ndet)


@pytest.mark.parametrize("signal", ["B2A", "B1C"])
def test_chain_in_one_native_call(ctx, signal, tmp_path):
    """bds_acquire_track: the same chain without returning to the host language between the stages (channel allocation by
    the device kernel); acqResults, channel and trackResults must be the bits of the three separate calls."""
    if signal == "B2A":
        s = bds_amd.init_settings_b2a(samplingFreq=25e6, IF=6.5e6, acqSatelliteList=[5, 9, 19, 33], acqSearchBand=2500,
                                      fineNoncoh=5, msToProcess=40, numberOfChannels=3, CNoInterval=20)
        sats = [synth.Sat(9, -1230.0, 12345.6, 2.0, 50.0), synth.Sat(19, 2210.0, 3001.2, 0.4, 47.0)]
        n_codes, acq_codes = 60, 8
    else:
        s = bds_amd.init_settings_b1c(samplingFreq=12.5e6, IF=3.5e6, acqSatelliteList=[3, 7, 12], acqSearchBand=600,
                                      msToProcess=60, numberOfChannels=3, pilotTRKflag=2, CNoInterval=3, FEBW=10e6)
        sats = [synth.Sat(3, 230.0, 40000.3, 1.0, 48.0), synth.Sat(12, -410.0, 99000.8, 2.0, 46.0)]
        n_codes, acq_codes = 10, 4
    spc = spc_of(s)
    x = synth.make_if(s, sats, n_codes * spc, seed=123)
    path = tmp_path / "record.bin"
    x.tofile(path)
    block = x[: acq_codes * spc]
    acq = bds_amd.acquisition(block, s, verbose=False)
    ch = bds_amd.pre_run(acq, s)
    trk, _ = bds_amd.tracking(str(path), ch, s)
    acq1, ch1, trk1 = bds_amd.acquire_track(block, str(path), s)
    for f in ("carrFreq", "codePhase", "peakMetric"):
        np.testing.assert_array_equal(getattr(acq1, f), getattr(acq, f))
    assert [_chan(c) for c in ch1] == [_chan(c) for c in ch]
    for a, b in zip(trk, trk1):
        assert a.status == b.status and a.PRN == b.PRN and a.completed == b.completed
        for f in ("absoluteSample", "I_P", "Q_P", "I_E", "I_L", "carrFreq", "codeFreq", "remCodePhase", "DataCNo"):
            np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)
