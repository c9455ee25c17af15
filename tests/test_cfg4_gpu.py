"""BASELINE.json configs[3] as ONE piece on the GPU: B1C wide-band tracking (BDS-3_B1C/WB_tracking.m:195-470), 12 channels x
36 000 ms = 3 600 ten-millisecond epochs at fs = 99.375 MS/s, from a raw int8 FILE of 3.6 GB (the reference's fid).

The record is made of 20-ms blocks that carry the 12 satellites seamlessly and one of 32 noise realisations each
(bench.cfg4_record), so the loops lock and size-independent properties hold over the whole run:
  * every channel completes all 3 600 epochs (status 'T') and the file is read to the expected position;
  * absoluteSample is the running sum of the block sizes the reference's own expression gives for the reported loop state:
    absoluteSample[k+1] - absoluteSample[k] == ceil((codeLength - remCodePhase[k]) / (codeFreq[k] / fs))  (WB_tracking.m:226-233;
    codeFreq[k] is stored before the epoch's update, :387-389);
  * 12 of 12 loops are locked over the second half and the C/N0 estimates average the injected 47 dB-Hz to within 1 dB;
  * the first 200 epochs of three channels equal the float64 oracle run on the head of the same record (tolerances of SURVEY.md section 8d:
    I/Q 1e-4 of |P|, carrFreq 1e-3 Hz, codeFreq 1e-6 Hz, absoluteSample exact; the strict default correlator has no ceil() flip, see
    tests/test_track_long_gpu.py)."""
import os

import numpy as np
import pytest

import bds_amd
import bench

pytestmark = pytest.mark.gpu
EPOCHS = 3600


def test_cfg4_twelve_channels_36_seconds_from_a_file(ctx, tmp_path):
    base = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1)
    s, ch, blocks, order, shift, n, spc = bench.cfg4_record(base, EPOCHS)
    path = os.path.join(os.environ.get("BDS_BENCH_TMP", str(tmp_path)), "cfg4.bin")
    bench.write_record(path, blocks, order, shift, n)
    try:
        assert os.path.getsize(path) == n and n > 3.5e9
        res, _ = bds_amd.tracking(path, ch, s, mode="WB")
        loaded = ctx.track_loaded_bytes() if hasattr(ctx, "track_loaded_bytes") else None
    finally:
        os.remove(path)
    assert len(res) == 12
    if loaded is not None:  # only the window the channels can touch travels to HBM: about the record, not more
        assert 0.9 * n < loaded <= n
    fs, code_len = s.samplingFreq, float(s.codeLength)
    half = EPOCHS // 2
    cnos = []
    for c, r in zip(ch, res):
        assert r.status == "T" and r.completed == EPOCHS
        # running sum of block sizes, from the loop state the call itself reports
        blk = np.ceil((code_len - r.remCodePhase) / (r.codeFreq / fs))
        assert r.absoluteSample[0] == c.codePhase - 1
        np.testing.assert_array_equal(np.diff(r.absoluteSample), blk[:-1])
        assert r.absoluteSample[-1] + blk[-1] <= n
        assert np.all(np.abs(blk - spc) <= 2)  # |Doppler| <= 1.5 kHz: a 10-ms block is 993 750 samples +- 1
        # locked: prompt energy sits in the in-phase arm of the data channel and of the QMBOC pilot
        assert np.abs(r.I_P[half:]).mean() > 3 * np.abs(r.Q_P[half:]).mean()
        assert np.abs(r.Pilot_I_P[half:]).mean() > 3 * np.abs(r.Pilot_Q_P[half:]).mean()
        # frequencies stay at the injected Doppler (50-Hz grid) and the record's mean code rate
        assert abs(np.mean(r.carrFreq[half:]) - c.acquiredFreq) < 2.0
        # (the record's code runs at the nominal rate -- whole code periods per 20-ms block --, and that is what the DLL
        #  settles on, pulling out the Doppler aiding preRun put into the channel's codeFreq)
        assert abs(np.mean(r.codeFreq[half:]) - s.codeFreqBasis) < 0.05, np.mean(r.codeFreq[half:])
        cnos.append(float(np.mean(r.B1C_CNo[len(r.B1C_CNo) // 2:])))
    # Calc_CNo_PLD (moments of 50 prompts, data + pilot): 44.3 .. 47.6 dB-Hz per channel on this record (the other 11
    # satellites are part of each channel's noise), 46.5 on average, for 47 injected
    assert all(abs(v - 47.0) < 3.5 for v in cnos), cnos
    assert abs(float(np.mean(cnos)) - 47.0) < 1.0, cnos


def _compare_with_oracle(ref, got, n_epochs):
    worst = dict(iq=0.0, carr=0.0, code=0.0)
    for r, g in zip(ref, got):
        assert g.status == "T" and r.status == "T"
        np.testing.assert_array_equal(g.absoluteSample[:n_epochs], r.absoluteSample[:n_epochs])
        p = np.hypot(r.I_P, r.Q_P).max()
        for f in ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_E", "Pilot_I_P", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_P", "Pilot_Q_L"):
            np.testing.assert_allclose(getattr(g, f)[:n_epochs], getattr(r, f)[:n_epochs], rtol=0, atol=1e-4 * p, err_msg=f)
            worst["iq"] = max(worst["iq"], float(np.max(np.abs(getattr(g, f)[:n_epochs] - getattr(r, f)[:n_epochs])) / p))
        np.testing.assert_allclose(g.carrFreq[:n_epochs], r.carrFreq[:n_epochs], rtol=0, atol=1e-3)
        np.testing.assert_allclose(g.codeFreq[:n_epochs], r.codeFreq[:n_epochs], rtol=0, atol=1e-6)
        worst["carr"] = max(worst["carr"], float(np.max(np.abs(g.carrFreq[:n_epochs] - r.carrFreq[:n_epochs]))))
        worst["code"] = max(worst["code"], float(np.max(np.abs(g.codeFreq[:n_epochs] - r.codeFreq[:n_epochs]))))
    return worst


def test_cfg4_head_against_the_oracle(ctx):
    """First 400 epochs (4 s of signal at 99.375 MS/s) of the cfg4 record, ALL 12 channels, vs the float64 oracle at SURVEY 8d: far
    beyond where the fp32 carrier of rounds 2-4 left the oracle's trajectory (its first ceil() flip came at epoch 142 on the long
    fixture); the strict default holds every epoch.  The oracle's sample loops run in C (oracle/c/trk_oracle.c, one thread per
    channel; held against the all-NumPy epoch by tests/test_oracle_c.py), its loop filters in oracle/tracking.py."""
    from oracle import cfast

    cfast.build()
    n_ep = 400
    base = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1)
    s, ch, blocks, order, shift, n, spc = bench.cfg4_record(base, n_ep)
    x = bench.record_bytes(blocks, order, shift, n)
    ref = cfast.tracking_parallel(x, ch, s, mode="WB")
    got, _ = bds_amd.tracking(x, ch, s, mode="WB")
    worst = _compare_with_oracle(ref, got, n_ep)
    print(f"cfg4 head, 12 channels x {n_ep} epochs vs the oracle: worst I/Q {worst['iq']:.2e} of |P|, carrFreq {worst['carr']:.2e} Hz, "
          f"codeFreq {worst['code']:.2e} Hz")


def _episodes(bad):
    """[(first, last)] runs of epochs outside the tolerance, runs less than 20 epochs apart merged (one loop transient)"""
    idx = np.nonzero(bad)[0]
    out = []
    for k in idx:
        if out and k - out[-1][1] <= 20:
            out[-1][1] = int(k)
        else:
            out.append([int(k), int(k)])
    return out


_SKIP_WH = pytest.mark.skipif(bool(os.environ.get("BDS_TEST_SKIP_WHOLE_HORIZON")), reason="BDS_TEST_SKIP_WHOLE_HORIZON set (the whole-horizon runs "
                              "take 1-2 min of host time each for the oracle)")
_OPT_IN = pytest.mark.skipif(not os.environ.get("BDS_TEST_CFG4_FULL"), reason="the reference's checked-in 53 MS/s settings on the same make of record: "
                             "BDS_TEST_CFG4_FULL=1 (the two 99.375 MS/s cases of BASELINE configs[3] always run)")


@pytest.mark.parametrize("mode,fs,epochs,nch", [pytest.param("WB", 99.375e6, EPOCHS, 12, marks=_SKIP_WH, id="cfg4-WB"),
                                                pytest.param("NB", 99.375e6, EPOCHS, 12, marks=_SKIP_WH, id="cfg4-NB"),
                                                pytest.param("WB", 53e6, 3700, 10, marks=_OPT_IN, id="b1c-defaults-WB")])
def test_cfg4_whole_horizon_against_the_c_oracle(ctx, tmp_path, mode, fs, epochs, nch):
    """BASELINE.json configs[3] literally: 12 channels x 36 000 ms at 99.375 MS/s from the 3.6 GB file, every epoch of every channel
    against the float64 oracle (sample loops in C, one thread per channel).  absoluteSample must be exact everywhere.  SURVEY 8d's
    closed-loop tolerances (I/Q 1e-4 of |P|, carrFreq 1e-3 Hz, codeFreq 1e-6 Hz) hold on every channel up to its first ceil() flip:
    the HIP path and the oracle agree to ~4e-10 of |P| and ~1e-12 chip, 43 200 epoch-channels of 1e6 samples x 21 code-index
    ceil()s each put ~2 samples within that distance of a chip boundary, ONE flipped sample is 2 |x| ~ 40 against a tolerance of
    1e-4 |P| ~ 20, and from there on the two loops -- both locked on the same satellite -- differ by a bounded floor that sustains
    itself (a code phase 1e-6 chip apart flips ~20 samples per epoch).  DESIGN.md section 2: any two float64 implementations that
    differ in a sin or a summation order do this to each other, MATLAB and NumPy included: the correlator sums of the HIP path are
    1e-13 of |P| from the oracle's, which moves codeFreq by an ulp now and then and remCodePhase with it by an ulp of the code length
    (1.8e-12 chip; tools/exp/r5_cfg4_flip.py).  Measured (profiles/r05_cfg4_full_vs_c_oracle.txt): with the default correlator
    (BDS_TRK_PREC=4) 11 of 12 channels inside 8d over all 3 600 epochs, one separates at epoch 1 530 on a single BOC(6,1) sample;
    with BDS_TRK_PREC=5 (4e-10 from the oracle) six separate between epochs 1 531 and 3 517.
    Asserted: few such separations, each starting from a single-sample-sized discrepancy, the floor after them bounded (I/Q 1e-2 of
    |P|, carrFreq 0.06 Hz, codeFreq 0.02 Hz: the bounds tests/test_track_long_gpu.py held the fp32 carrier to); 8d everywhere else."""
    from oracle import cfast

    cfast.build()
    # ("b1c-defaults": the reference's own checked-in B1C settings -- B1C/initSettings.m:57,62,67: fs = 53 MS/s, msToProcess = 37 000,
    #  numberOfChannels = 10, pilotTRKflag = 2 -- on a record of the same make)
    base = bds_amd.init_settings_b1c(samplingFreq=fs, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1)
    EPOCHS = epochs  # noqa: N806  (shadows the module constant for this run)
    s, ch, blocks, order, shift, n, spc = bench.cfg4_record(base, EPOCHS)
    if nch != 12:
        ch, s = ch[:nch], s.copy(numberOfChannels=nch)
    if mode == "NB":  # NB_tracking.m on the same record (B1C/postProcessing.m:137-143 picks the variant by pilotTRKflag)
        s = s.copy(pilotTRKflag=1)
    path = os.path.join(os.environ.get("BDS_BENCH_TMP", str(tmp_path)), "cfg4_full.bin")
    bench.write_record(path, blocks, order, shift, n)
    try:
        got, _ = bds_amd.tracking(path, ch, s, mode=mode)
        data = np.memmap(path, dtype=np.int8, mode="r")
        ref = cfast.tracking_parallel(data, ch, s, mode=mode)
        del data
    finally:
        os.remove(path)
    iq_fields = ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_P", "Pilot_Q_P") + (
        ("Pilot_I_E", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_L") if mode == "WB" else ())
    n_bad = n_eps = 0
    quiet = dict(iq=0.0, carr=0.0, code=0.0)
    for c, (r, g) in enumerate(zip(ref, got)):
        assert g.status == "T" and r.status == "T"
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)  # every block size of every epoch
        p = np.hypot(r.I_P, r.Q_P).max()
        d_iq = np.max(np.stack([np.abs(getattr(g, f) - getattr(r, f)) for f in iq_fields]), axis=0) / p
        d_carr, d_code = np.abs(g.carrFreq - r.carrFreq), np.abs(g.codeFreq - r.codeFreq)
        bad = (d_iq > 1e-4) | (d_carr > 1e-3) | (d_code > 1e-6)
        eps = _episodes(bad)
        for a, b in eps:
            print(f"  channel {c} (PRN {r.PRN}): epochs {a + 1}..{b + 1} outside 8d ({b - a + 1} epochs): first discrepancy {d_iq[a] * p:.1f} "
                  f"(= {d_iq[a]:.2e} of |P|), worst I/Q {d_iq[a:b + 1].max():.2e} of |P|, carrFreq {d_carr[a:b + 1].max():.2e} Hz, "
                  f"codeFreq {d_code[a:b + 1].max():.2e} Hz")
            assert d_iq[a] * p <= 4 * 127 * 2          # starts with a few samples' worth, not with a wrong trajectory
            assert d_iq[a:b + 1].max() <= 1e-2 and d_code[a:b + 1].max() <= 0.02 and d_carr[a:b + 1].max() <= 0.06
        n_bad += int(bad.sum())
        n_eps += len(eps)
        ok = ~bad
        quiet["iq"] = max(quiet["iq"], float(d_iq[ok].max()))
        quiet["carr"] = max(quiet["carr"], float(d_carr[ok].max()))
        quiet["code"] = max(quiet["code"], float(d_code[ok].max()))
    total = EPOCHS * len(ref)
    print(f"whole horizon ({mode}_tracking), {len(ref)} channels x {EPOCHS} epochs x {fs / 1e6:g} MS/s vs the oracle: absoluteSample exact on all {total} epoch-channels; "
          f"{total - n_bad} inside SURVEY 8d (worst there: I/Q {quiet['iq']:.2e} of |P|, carrFreq {quiet['carr']:.2e} Hz, codeFreq {quiet['code']:.2e} Hz); "
          f"{n_eps} separation(s) after a ceil() flip, {n_bad} epoch-channels after them")
    assert n_eps <= 8 and n_bad <= 0.25 * total
