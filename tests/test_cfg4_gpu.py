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


def test_cfg4_head_against_the_oracle(ctx):
    """First 200 epochs (2 s of signal at 99.375 MS/s) of the cfg4 record, three of its channels, vs the float64 oracle at SURVEY 8d:
    far beyond where the fp32 carrier of rounds 2-4 left the oracle's trajectory (its first ceil() flip came at epoch 142 on the
    long fixture); the strict default holds every epoch (~80 s of oracle time)."""
    from oracle import tracking as otrk

    base = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1)
    s, ch, blocks, order, shift, n, spc = bench.cfg4_record(base, 200)
    x = bench.record_bytes(blocks, order, shift, n)
    sub = [ch[0], ch[5], ch[11]]
    ref, _ = otrk.tracking(otrk.RawFile(x), sub, s.copy(numberOfChannels=3), mode="WB")
    got, _ = bds_amd.tracking(x, sub, s.copy(numberOfChannels=3), mode="WB")
    for r, g in zip(ref, got):
        assert g.status == "T"
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)
        p = np.hypot(r.I_P, r.Q_P).max()
        for f in ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_E", "Pilot_I_P", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_P", "Pilot_Q_L"):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-4 * p, err_msg=f)
        np.testing.assert_allclose(g.carrFreq, r.carrFreq, rtol=0, atol=1e-3)
        np.testing.assert_allclose(g.codeFreq, r.codeFreq, rtol=0, atol=1e-6)
