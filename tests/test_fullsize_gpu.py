"""BASELINE.json full-size acquisition configs on the GPU (configs[1] B2a and configs[2] B1C, 63 PRNs,
fs = 99.375 MS/s): size-independent properties + oracle spot checks on a handful of cells (the full
float64 oracle would need ~1 h for the B1C grid)."""
import numpy as np
import pytest

import bds_amd
from oracle import acquisition as oacq

import bench

pytestmark = pytest.mark.gpu


def _check_injected(s, sats, res, spc, thr, ftol):
    injected = {sat.prn: sat for sat in sats}
    detected = set(int(p) for p in np.nonzero(res.carrFreq)[0] + 1)
    assert detected == set(injected)
    for prn, sat in injected.items():
        assert abs(res.carrFreq[prn - 1] - (s.IF + sat.doppler)) <= ftol  # estimator noise at 45 dB-Hz, not parity
        d = ((res.codePhase[prn - 1] - 1) - sat.delay) % spc
        assert min(d, spc - d) <= 6.0  # code-delay estimate, noise limited (a half-chip is 48.6 samples)
        assert res.peakMetric[prn - 1] > thr
    for prn in s.acqSatelliteList:
        if prn not in injected:
            assert 0 < res.peakMetric[prn - 1] <= thr and res.codePhase[prn - 1] == 0


def test_b2a_full_grid(ctx):
    s, x, sats, _ = bench.build_workload("b2a")
    res = bds_amd.acquisition(x, s, verbose=False)
    _check_injected(s, sats, res, 99375, s.acqThreshold, 100.0)
    # oracle on three PRNs (two present, one absent), whole Doppler grid
    sub = s.copy(acqSatelliteList=[4, 19, 33])
    ref = oacq.acquisition_b2a(x.astype(np.float64), sub)
    for p in (4, 19, 33):
        assert res.codePhase[p - 1] == ref.codePhase[p - 1] and res.carrFreq[p - 1] == ref.carrFreq[p - 1]
        np.testing.assert_allclose(res.peakMetric[p - 1], ref.peakMetric[p - 1], rtol=1e-6)


def test_b1c_full_grid(ctx):
    s, x, sats, _ = bench.build_workload("b1c")
    res = bds_amd.acquisition(x, s, verbose=False)
    _check_injected(s, sats, res, 993750, s.acqThreshold, 50.0)
    tm = ctx.timing()
    assert tm["n_bins"] == 201 and tm["n_prn"] == 63 and tm["n_circ"] == 1987500
    rm, ra = ctx.acq_grid(63, 201)
    pk, dn, fb = ctx.acq_peaks(63)
    tol = {0: 1e-5, 1: 1e-3}[tm["half_storage"]]  # GRID_TOL: kDelta / 2 with fp32 storage, HALF of kDelta / 2 = 2e-3 with fp16 storage (AcqRun::setup: kDelta = 4e-3)
    xf = x.astype(np.float64)
    # oracle rows of the winning bin and its neighbours for one present and one absent PRN
    for prn in (sats[0].prn, 2):
        b = int(fb[prn - 1]) - 1
        bins = [bb for bb in (b - 1, b, b + 1) if 0 <= bb < 201]
        best = -1.0
        for bb, row in oacq.b1c_coarse_rows(xf, s, prn, bins):
            np.testing.assert_allclose(rm[prn - 1, bb], row.max(), rtol=tol)
            best = max(best, row.max())
            if bb == b:
                assert int(np.argmax(row)) + 1 == int(ra[prn - 1, bb]) or tm["half_storage"]
        np.testing.assert_allclose(pk[prn - 1], best, rtol=1e-9)  # f64 refinement == oracle peak
        sig_power = np.sqrt(np.var(xf[:993750], ddof=1) * 993750)
        np.testing.assert_allclose(res.peakMetric[prn - 1], best / sig_power, rtol=1e-9)
    # sharding invariance at full size: two halves sum to the full result
    a = bds_amd.acquisition(x, s, prn_list=list(range(1, 64, 2)), verbose=False)
    b2 = bds_amd.acquisition(x, s, prn_list=list(range(2, 64, 2)), verbose=False)
    for f in ("carrFreq", "codePhase", "peakMetric"):
        np.testing.assert_array_equal(getattr(a, f) + getattr(b2, f), getattr(res, f))


def test_b1c_full_grid_absent_prn_against_the_whole_oracle_matrix(ctx):
    """BASELINE.json configs[2], one PRN that is NOT in the block, against the oracle on ALL 201 Doppler bins
    (201 x 1 987 500 results matrix, B1C/acquisition.m:191-222): the global argmax of a noise-only surface is the
    hard case for the sieve -- no peak stands out, so a wrong tile record or a hidden near-tie would change
    fbin / codePhase.  max(max(results)) (:229-235), its bin and its code phase must match exactly."""
    s, x, sats, _ = bench.build_workload("b1c")
    prn = 2
    assert prn not in [sat.prn for sat in sats]
    sub = s.copy(acqSatelliteList=[prn])
    res = bds_amd.acquisition(x, sub, verbose=False)
    tm = ctx.timing()
    pk, dn, fb = ctx.acq_peaks(63)
    rm, ra = ctx.acq_grid(1, 201)
    cand = set(map(tuple, ctx.acq_candidates(prn).tolist()))
    xf = x.astype(np.float64)
    best, best_b, best_lag = -1.0, -1, -1
    tol = {0: 1e-5, 1: 1e-3}[tm["half_storage"]]  # fp32 storage: kDelta / 2; fp16 storage: half of kDelta / 2 = 2e-3 (kDelta = 4e-3)
    near = []
    for b, row in oacq.b1c_coarse_rows(xf, sub, prn):
        m = float(row.max())
        np.testing.assert_allclose(rm[0, b], m, rtol=tol)  # the sieve's row maximum, within half its tolerance
        if m > best:  # first maximal row / first maximal column, like max(max(results))
            best, best_b, best_lag = m, b, int(np.argmax(row))
        near.append((b, row))
        near[-1] = (b, np.nonzero(row >= (1 - tol) * m)[0], row[row >= (1 - tol) * m])
    assert fb[prn - 1] == best_b + 1
    np.testing.assert_allclose(pk[prn - 1], best, rtol=1e-9)
    assert (best_b + 1, best_lag + 1) in cand
    # every cell of the whole matrix within kDelta / 2 of the global maximum was refined
    for b, lags, vals in near:
        for lag, v in zip(lags.tolist(), vals.tolist()):
            if v >= (1 - tol) * best:
                assert (b + 1, lag + 1) in cand, (b, lag, v, best)
    # acqResults of the absent PRN: metric below threshold -> zeros, metric = peak / sigPower
    sig_power = np.sqrt(np.var(xf[:993750], ddof=1) * 993750)
    np.testing.assert_allclose(res.peakMetric[prn - 1], best / sig_power, rtol=1e-9)
    assert res.carrFreq[prn - 1] == 0 and res.codePhase[prn - 1] == 0


def _usable_cpus():
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(np.ceil(float(q) / float(per)))))
    except Exception:  # noqa: BLE001
        pass
    return n


def test_b1c_full_grid_every_prn_against_the_c_oracle(ctx):
    """BASELINE.json configs[2] against the oracle on the WHOLE Doppler grid of EVERY PRN (63 x 201 cells x 1 987 500 lags; ~2.5 min
    on the 16 CPUs of the pool's GPU boxes, profiles/r05_cfg3_all_prns_vs_c_oracle.txt; BDS_TEST_FEW_PRNS=1 keeps the ten injected
    satellites and six absent PRNs).  The rows come from the
    compiled restatement (oracle/c/acq_oracle.c, OpenMP over the bins: seconds per PRN where the NumPy rows take minutes; the two
    are held together by tests/test_oracle_c.py), everything after the rows from the NumPy oracle (B1C/acquisition.m:229-307).
    Per PRN: all 201 sieve row maxima within 1e-3 (half of kDelta / 2 at fp16 storage), the f64 peak to 1e-9, bin / codePhase / carrFreq exact, peakMetric 1e-9."""
    import os

    from oracle import cfast

    cfast.build()
    s, x, sats, _ = bench.build_workload("b1c")
    res = bds_amd.acquisition(x, s, verbose=False)
    tm = ctx.timing()
    rm, ra = ctx.acq_grid(63, 201)
    pk, dn, fb = ctx.acq_peaks(63)
    tol = {0: 1e-5, 1: 1e-3}[tm["half_storage"]]  # fp32 storage: kDelta / 2; fp16 storage: half of kDelta / 2 = 2e-3 (kDelta = 4e-3)
    present = [sat.prn for sat in sats]
    prns = present + [2, 3, 30, 45, 60, 63] if os.environ.get("BDS_TEST_FEW_PRNS") else list(range(1, 64))
    xf = x[:3 * 993750 + 16].astype(np.float64)  # (acquisition touches N + spc - 1 samples)
    rows = {}

    def coarse(long_signal, settings, prn):
        out = cfast.coarse_rows(long_signal, settings, prn, threads=_usable_cpus())
        rows[prn] = out
        return out[0], out[1], out[2], None

    ref = oacq.acquisition_b1c(xf, s.copy(acqSatelliteList=prns), coarse=coarse)
    worst = 0.0
    for prn in prns:
        row_max, row_arg, col_max, _ = rows[prn]
        np.testing.assert_allclose(rm[prn - 1], row_max, rtol=tol)
        worst = max(worst, float(np.max(np.abs(rm[prn - 1] - row_max)) / row_max.max()))
        assert fb[prn - 1] == int(np.argmax(row_max)) + 1
        np.testing.assert_allclose(pk[prn - 1], col_max.max(), rtol=1e-9)
        assert res.codePhase[prn - 1] == ref.codePhase[prn - 1] and res.carrFreq[prn - 1] == ref.carrFreq[prn - 1], prn
        np.testing.assert_allclose(res.peakMetric[prn - 1], ref.peakMetric[prn - 1], rtol=1e-9)
        assert (ref.carrFreq[prn - 1] != 0) == (prn in present)
    print(f"cfg3 vs the C oracle: {len(prns)} PRNs x 201 bins, worst sieve row maximum error {worst:.3e} of the PRN maximum "
          f"(asserted: {tol:g}; kDelta / 2 = {2 * tol if tol > 1e-4 else tol:g})")


def test_b2a_full_grid_every_prn_against_the_c_oracle(ctx):
    """BASELINE.json configs[1], ALL 63 PRNs x 26 bins against the oracle (rows from the compiled restatement, the second peak,
    the threshold and the fine search of B2a/acquisition.m:213-336 from the NumPy oracle): acqResults exact / 1e-6."""
    from oracle import cfast

    cfast.build()
    s, x, sats, _ = bench.build_workload("b2a")
    res = bds_amd.acquisition(x, s, verbose=False)
    ref = oacq.acquisition_b2a(x.astype(np.float64), s, coarse=cfast.backend(threads=_usable_cpus()))
    np.testing.assert_array_equal(res.codePhase, ref.codePhase)
    np.testing.assert_array_equal(res.carrFreq, ref.carrFreq)
    np.testing.assert_allclose(res.peakMetric, ref.peakMetric, rtol=1e-6)
    assert set(np.nonzero(ref.carrFreq)[0] + 1) == {sat.prn for sat in sats}


@pytest.mark.parametrize("fs,iq", [(53e6, False), (30.69e6, False), (53e6, True)], ids=["53MSps", "30.69MSps", "53MSps-IQ"])
def test_b1c_at_the_references_own_sampling_rates_against_the_c_oracle(ctx, fs, iq):
    """The configuration a user of the reference runs first: B1C/initSettings.m as checked in -- fs = 53 MS/s (:57; N = 1 060 000 =
    2^5 5^4 53), IF = 14.58 MHz, acqSatelliteList = [19 20], 201 bins, 10 ms data + pilot -- and the rate its comment keeps beside it
    (30.69 MS/s: N = 613 800 = 2^3 3^2 5^2 11 31), at full size against the oracle on the whole grid (rows from the C restatement,
    whose transform takes any length; the HIP path pads to its own 5-smooth two-pass length either way).  The third case is the same
    configuration on a fileType-2 record (interleaved I/Q int8, B1C/postProcessing.m:92-96): complex longSignal end to end."""
    from oracle import cfast

    from bds_amd import synth
    from helpers import spc_of

    cfast.build()
    s = bds_amd.init_settings_b1c(samplingFreq=fs, fileType=2 if iq else 1)
    assert s.acqSatelliteList == [19, 20] and s.IF == 1590e6 - 1575.42e6
    spc = spc_of(s)
    sats = [synth.Sat(19, -1730.0, 0.613 * spc, 0.7, 45.0), synth.Sat(35, 2210.0, 0.2 * spc, 2.0, 46.0)]  # PRN 20 absent, PRN 35 not searched
    x = synth.make_if(s, sats, 4 * spc, seed=53, iq_sign=-1 if iq else 0)
    if iq:
        from helpers import as_complex

        x = as_complex(x)
    res = bds_amd.acquisition(x, s, verbose=False)
    tm = ctx.timing()
    assert tm["n_bins"] == 201 and tm["n_circ"] == 2 * spc
    rm, ra = ctx.acq_grid(2, 201)
    pk, dn, fb = ctx.acq_peaks(20)
    tol = {0: 1e-5, 1: 1e-3}[tm["half_storage"]]
    rows = {}

    def coarse(long_signal, settings, prn):
        out = cfast.coarse_rows(long_signal, settings, prn, threads=_usable_cpus())
        rows[prn] = out
        return out[0], out[1], out[2], None

    ref = oacq.acquisition_b1c(x if iq else x.astype(np.float64), s, coarse=coarse)
    for i, prn in enumerate((19, 20)):
        row_max, row_arg, col_max, _ = rows[prn]
        np.testing.assert_allclose(rm[i], row_max, rtol=tol)
        assert fb[prn - 1] == int(np.argmax(row_max)) + 1
        np.testing.assert_allclose(pk[prn - 1], col_max.max(), rtol=1e-9)
    np.testing.assert_array_equal(res.codePhase, ref.codePhase)
    np.testing.assert_array_equal(res.carrFreq, ref.carrFreq)
    np.testing.assert_allclose(res.peakMetric, ref.peakMetric, rtol=1e-9)
    assert ref.carrFreq[18] != 0 and ref.carrFreq[19] == 0
    assert abs(ref.carrFreq[18] - (s.IF - 1730.0)) <= 25 and abs((ref.codePhase[18] - 1) - 0.613 * spc) % spc <= 3


@pytest.mark.parametrize("name", ["b2a", "b1c"])
def test_resampling_branch_at_full_size_against_the_c_oracle(ctx, name):
    """The reference's own speed-up option at the BASELINE sizes: resamplingflag = 1 on the cfg2 / cfg3 blocks (fs = 99.375 MS/s above
    either receiver's resamplingThreshold) -- fir1(700) + filtfilt + band-pass-sampling decimation to 48.06 MS/s (B2a) / 19.62 MS/s (B1C:
    N shrinks from 1 987 500 to 392 400), the search on the conditioned block, codePhase / carrFreq mapped back
    (B2a/acquisition.m:54-124,339-356; B1C/acquisition.m:54-123,311-328).  Whole Doppler grids against the oracle (scipy's firwin /
    filtfilt for the conditioner, rows from the C restatement): B2a all 63 PRNs, B1C the ten injected satellites and six absent PRNs
    (all 63 with BDS_TEST_ALL_PRNS=1)."""
    import os

    from oracle import cfast

    cfast.build()
    s, x, sats, _ = bench.build_workload(name)
    present = [sat.prn for sat in sats]
    if name == "b1c" and not os.environ.get("BDS_TEST_ALL_PRNS"):
        s = s.copy(acqSatelliteList=present + [2, 3, 30, 45, 60, 63])
        x = x[:6 * 993750]  # (filtfilt runs over the whole longSignal: keep the oracle's share of it short)
    s = s.copy(resamplingflag=1)
    res = bds_amd.acquisition(x, s, verbose=False)
    fn = oacq.acquisition_b1c if name == "b1c" else oacq.acquisition_b2a
    ref = fn(x.astype(np.float64), s, coarse=cfast.backend(threads=_usable_cpus()))
    np.testing.assert_array_equal(res.codePhase, ref.codePhase)
    np.testing.assert_array_equal(res.carrFreq, ref.carrFreq)
    np.testing.assert_allclose(res.peakMetric, ref.peakMetric, rtol=1e-6)
    det = set(int(p) for p in np.nonzero(ref.carrFreq)[0] + 1)
    assert det and det <= set(present)  # (the decimated block is noisier: a weak satellite may drop below the threshold -- in both)


def test_argument_errors_are_reported(ctx):
    s = bds_amd.init_settings_b2a(acqSatelliteList=[5])
    x = np.zeros(1000, dtype=np.int8)
    with pytest.raises(bds_amd.native.BdsError, match="acquisition needs at least"):
        bds_amd.acquisition(x, s, verbose=False)
    x = np.zeros(17 * 99375, dtype=np.int8)
    with pytest.raises(bds_amd.native.BdsError, match="resampling band edges"):  # fir1 would reject them
        bds_amd.acquisition(x, s.copy(resamplingflag=1, IF=5e6), verbose=False)
    with pytest.raises(bds_amd.native.BdsError, match="out of 1..63"):
        bds_amd.acquisition(x, s.copy(acqSatelliteList=[64]), verbose=False)
    with pytest.raises(bds_amd.native.BdsError, match="acquisition needs at least"):
        bds_amd.acquisition(x[:1000].astype(np.complex128), s.copy(fileType=2), verbose=False)
    with pytest.raises(bds_amd.native.BdsError, match="acqStep"):
        bds_amd.acquisition(x, s.copy(acqStep=0), verbose=False)


def test_all_zero_block_is_not_a_detection(ctx):
    """Degenerate input: every correlation is 0, the B2a ratio is 0/0 -> NaN, NaN > threshold is false."""
    s = bds_amd.init_settings_b2a(samplingFreq=25e6, IF=6.5e6, acqSatelliteList=[5, 9], acqSearchBand=800)
    x = np.zeros(8 * 25000, dtype=np.int8)
    r = bds_amd.acquisition(x, s, verbose=False)
    assert not np.any(r.carrFreq) and not np.any(r.codePhase)


def test_b1c_wideband_tracking_full_rate(ctx):
    """BASELINE.json configs[3] shape, shortened: B1C wide-band tracking, 12 channels, fs = 99.375 MS/s,
    3 closed-loop 10-ms epochs vs the oracle (tolerances of SURVEY.md section 8d)."""
    from types import SimpleNamespace

    from bds_amd import synth
    from oracle import tracking as otrk

    n_epochs = 3
    s = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, msToProcess=n_epochs * 10, numberOfChannels=12,
                                  pilotTRKflag=2, CNoInterval=50)
    spc = 993750
    rng = np.random.default_rng(44)
    prns = [1, 4, 9, 14, 19, 20, 27, 35, 46, 58, 60, 63]
    sats = synth.random_sats(rng, prns, spc, cn0_dbhz=47.0)
    x = synth.make_if(s, sats, (n_epochs + 2) * spc, seed=45)
    chans = []
    for sat in sats:
        cf = s.IF + round(sat.doppler / 25) * 25
        chans.append(SimpleNamespace(PRN=sat.prn, acquiredFreq=float(cf), codePhase=float(int(np.ceil(sat.delay)) + 1),
                                     codeFreq=float(s.codeFreqBasis - (cf - s.IF) / s.carrFreqBasis * s.codeFreqBasis),
                                     status="T"))
    ref, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode="WB")
    got, _ = bds_amd.tracking(x, chans, s, mode="WB")
    for r, g in zip(ref, got):
        assert g.status == "T"
        np.testing.assert_array_equal(g.absoluteSample, r.absoluteSample)
        p = np.hypot(r.I_P, r.Q_P).max()
        for f in ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_E", "Pilot_I_P", "Pilot_I_L", "Pilot_Q_E",
                  "Pilot_Q_P", "Pilot_Q_L"):
            np.testing.assert_allclose(getattr(g, f), getattr(r, f), rtol=0, atol=1e-4 * p, err_msg=f)
        np.testing.assert_allclose(g.carrFreq, r.carrFreq, rtol=0, atol=1e-3)
        np.testing.assert_allclose(g.codeFreq, r.codeFreq, rtol=0, atol=1e-6)


_L = {"BDS_ACQ_PFA": "0"}  # the L-point pair of rounds 3-5 (768 x 4096) instead of the N-point pair (round 6: the default at cfg3)


@pytest.mark.parametrize("env", [_L, {**_L, "BDS_ACQ_PK": "0"}, {**_L, "BDS_ACQ_PK": "2"}, {**_L, "BDS_ACQ_ILV": "0"}, {**_L, "BDS_ACQ_PK": "0", "BDS_ACQ_ILV": "0"},
                                 {**_L, "BDS_ACQ_NEIGH": "1"}, {**_L, "BDS_ACQ_WROWS": "0"}, {**_L, "BDS_ACQ_WCOLS": "0"}, {**_L, "BDS_ACQ_HOSTREFINE": "1"},
                                 {**_L, "BDS_ACQ_PAIR_GB": "auto"}, {**_L, "BDS_ACQ_PAIR_GB": "11"}, {**_L, "BDS_ACQ_PAIR_GB": "auto", "BDS_ACQ_WCOLS": "0"},
                                 {"BDS_ACQ_HOSTREFINE": "1"}, {"BDS_ACQ_NEIGH": "1"}, {"BDS_ACQ_PAIR_GB": "auto"}, {"BDS_ACQ_PAIR_GB": "11"},
                                 {"BDS_ACQ_PFA_QCHUNK": "196"}, {"BDS_ACQ_PFA_CGRID": "1024"}])
def test_cfg3_kernel_variants_decide_the_same(env, monkeypatch):
    """The default at cfg3 is the N-point pair (csrc/bds_acq_pfa.h, round 6); BDS_ACQ_PFA=0 selects the L-point pair of rounds 3-5, whose
    own switches follow.  The N-point pair's variants: the refinement through the host, the +-1 neighbours, the serving mode / 11 GiB,
    a cell-major work list of its column pass, a quarter of its grid.
    The switches of the round-4 search kernels at the cfg3 plan (768 x 4096): plain-fp32 instead of packed butterflies (both
    passes / column pass only), component planes instead of interleaved components, the +-1 neighbours of rounds 1-3 refined
    as well, the round-2 row / column kernels, the refinement through the host (rounds 1-4) instead of the device chain of
    round 5 (csrc/bds_acq_refine.h), the serving mode of the search (BDS_ACQ_PAIR_GB, a release knob: all four PRNs' Doppler rows in
    one launch pair / two PRNs per pair at 11 GiB / with the tile column pass).  Every variant is a different sieve in front of the same f64 decision: acqResults
    must be the default's bit for bit, the search grid within the sieve's tolerance of it."""
    s, x, sats, _ = bench.build_workload("b1c")
    prns = [sats[0].prn, 2, sats[1].prn, 33]
    ref = bds_amd.native.Context(0)
    ref.acq_load(s, x)
    ref.acq_prepare(s)
    want = ref.acq_run(s, prn_list=prns)
    g0, _ = ref.acq_grid(len(prns), 201)
    flags0 = ref.timing()["kernel_flags"]
    ref.close()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = bds_amd.native.Context(0)  # the knobs are read once, at context creation
    c.acq_load(s, x)
    c.acq_prepare(s)
    got = c.acq_run(s, prn_list=prns)
    g1, _ = c.acq_grid(len(prns), 201)
    tm = c.timing()
    c.close()
    for u, v in zip(want, got):
        assert np.array_equal(u, v)
    np.testing.assert_allclose(g1, g0, rtol=2e-3)
    assert flags0 == 3  # default: interleaved components + packed butterflies
    if "BDS_ACQ_PFA" in env:
        assert tm["fft_len"] == 3145728 and tm["rows_kernel"] in (1, 2) and tm["cols_kernel"] in (1, 2)
    else:
        assert (tm["rows_kernel"], tm["cols_kernel"], tm["fft_len"]) == (3, 4, 1987500)
    if "BDS_ACQ_PAIR_GB" in env:
        assert tm["n_pairs"] == (1 if env["BDS_ACQ_PAIR_GB"] == "auto" else 2) and tm["cells_per_pair"] == 201 * 4 / tm["n_pairs"]
    else:
        assert tm["n_pairs"] == 4 and tm["cells_per_pair"] == 201
    assert tm["refine_path"] == (0 if set(env) & {"BDS_ACQ_NEIGH", "BDS_ACQ_WCOLS", "BDS_ACQ_HOSTREFINE"} else 1)
    if "BDS_ACQ_PK" in env and env["BDS_ACQ_PK"] == "0":
        assert not tm["kernel_flags"] & 2
    if "BDS_ACQ_ILV" in env:
        assert not tm["kernel_flags"] & 1
    if "BDS_ACQ_WROWS" in env:
        assert tm["rows_kernel"] == 1 and tm["kernel_flags"] == 0
    if "BDS_ACQ_WCOLS" in env:
        assert tm["cols_kernel"] == 1


def test_serving_mode_through_the_api(ctx):
    """bds_acq_set_pair_budget_gb (the API form of BDS_ACQ_PAIR_GB): the same context searches cfg3's first six PRNs all in one launch
    pair (the default budget of 40 GiB holds eight PRNs' cells; "auto" likewise), three per pair (16 GiB), one PRN per pair (0: the
    minimal footprint) -- the same bits every time."""
    s, x, sats, _ = bench.build_workload("b1c")
    prns = [1, 2, 3, 4, 5, 6]
    c = bds_amd.native.Context(0)
    try:
        c.acq_load(s, x)
        c.acq_prepare(s)
        out = []
        for budget, pairs in ((None, 1), (0, 6), ("auto", 1), (16, 2), (0, 6)):
            if budget is not None:
                c.acq_set_pair_budget(budget)
            res = c.acq_run(s, prn_list=prns)
            tm = c.timing()
            assert tm["n_pairs"] == pairs and abs(tm["cells_per_pair"] - 201 * 6 / pairs) < 1e-9, (budget, tm["n_pairs"], tm["cells_per_pair"])
            out.append(res)
    finally:
        c.close()
    for res in out[1:]:
        for u, v in zip(out[0], res):
            assert np.array_equal(u, v)
    assert out[0][0][0] != 0 and out[0][0][3] != 0 and out[0][0][1] == 0  # PRNs 1 and 4 are in the block, PRN 2 is not


def test_cfg2_refinement_paths_decide_the_same(monkeypatch):
    """cfg2 with the refinement as one device chain (default: candidates, f64 sums, peak, second-peak pass, threshold and fine
    search without a host round trip) and through the host (BDS_ACQ_HOSTREFINE=1, rounds 1-4): acqResults, the f64 peaks, the
    second peaks and the refined candidate cells are the same."""
    s, x, sats, _ = bench.build_workload("b2a")
    monkeypatch.setenv("BDS_VERBOSE", "1")
    out = {}
    # "0": the device chain, its second-peak pass reading the winning cells out of the main search's inter-pass buffer (round 5);
    # "2": the device chain with a row pass of its own for those cells (BDS_ACQ_NO_BWREUSE=1); "1": the host path
    for host, var in (("0", None), ("2", "BDS_ACQ_NO_BWREUSE"), ("1", "BDS_ACQ_HOSTREFINE")):
        monkeypatch.delenv("BDS_ACQ_NO_BWREUSE", raising=False)
        if var:
            monkeypatch.setenv(var, "1")
        c = bds_amd.native.Context(0)
        c.acq_load(s, x)
        c.acq_prepare(s)
        res = c.acq_run(s)
        tm = c.timing()
        out[host] = (res, c.acq_peaks(63), [c.acq_candidates(p) for p in (1, 19, 63)])
        c.close()
        assert tm["refine_path"] == (0 if host == "1" else 1)
    for other in ("1", "2"):
        for u, v in zip(out["0"][0], out[other][0]):
            assert np.array_equal(u, v)
        for u, v in zip(out["0"][1], out[other][1]):
            assert np.array_equal(u, v)
        for u, v in zip(out["0"][2], out[other][2]):
            assert len(u) > 0 and np.array_equal(u, v)


def test_cfg2_plans_decide_the_same(monkeypatch):
    """cfg2 (B2a, 63 PRNs x 26 bins) on the 80 x 4096 plan of round 4 (wave-private row pass + one lane per column,
    bds_acq_scols.h) and on the 256 x 1280 plan of rounds 1-3 (BDS_ACQ_SMALL=0): different sieves, the same f64 decision --
    acqResults bit for bit, the second peak included (peakMetric = peak / second peak)."""
    s, x, sats, _ = bench.build_workload("b2a")
    out = {}
    for small in ("1", "0"):
        monkeypatch.setenv("BDS_ACQ_SMALL", small)
        c = bds_amd.native.Context(0)
        c.acq_load(s, x)
        c.acq_prepare(s)
        out[small] = c.acq_run(s)
        tm = c.timing()
        c.close()
        assert (tm["plan_l1"], tm["plan_l2"], tm["cols_kernel"]) == ((80, 4096, 3) if small == "1" else (256, 1280, 1))
    for u, v in zip(out["1"], out["0"]):
        assert np.array_equal(u, v)
    assert sorted(int(p) for p in np.nonzero(out["1"][0])[0] + 1) == sorted(sat.prn for sat in sats)
