"""The N-point search pair (csrc/bds_acq_pfa.h: B1C at N = 1 987 500 = 53 x 12 x 3125, the default of BASELINE configs[2]) beyond the
bench block: which settings take it, and that every one of them decides what the L-point pair of rounds 3-5 (BDS_ACQ_PFA=0) and the
oracle decide.  The whole-grid comparisons with the oracle at cfg3 live in tests/test_fullsize_gpu.py (they run on this pair by default)."""
import numpy as np
import pytest

import bds_amd
import bench
from bds_amd import synth

pytestmark = pytest.mark.gpu
N = 1987500


def _run(monkeypatch, s, x, prns, env=None):
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    c = bds_amd.native.Context(0)  # the knobs are read once, at context creation
    for k in (env or {}):
        monkeypatch.delenv(k)
    try:
        c.acq_load(s, x)
        c.acq_prepare(s)
        res = c.acq_run(s, prn_list=prns)
        tm = c.timing()
        grid, arg = c.acq_grid(len(prns), int(tm["n_bins"]))
        pk, dn, fb = c.acq_peaks(63)
    finally:
        c.close()
    return res, tm, grid, arg, pk, fb


def _same_decisions(a, b):
    for u, v in zip(a[0], b[0]):
        assert np.array_equal(u, v)          # carrFreq, codePhase, peakMetric (f64 decisions): bit for bit
    np.testing.assert_array_equal(a[4], b[4])  # f64 peaks
    np.testing.assert_array_equal(a[5], b[5])  # winning bins
    np.testing.assert_allclose(a[2], b[2], rtol=2e-3)  # the two sieves' row maxima: within kDelta / 2 of each other


@pytest.fixture(scope="module")
def block():
    return bench.build_workload("b1c")


def test_doppler_steps_of_two_bins_and_a_narrow_band(block, monkeypatch):
    """acqStep = 100 Hz is two spectrum bins per Doppler bin (acqStep N / fs = 2), a +-4.1 kHz band moves the first bin's frequency: the
    rotation per bin and the one forward transform follow; the L-point pair transforms every bin on its own"""
    s0, x, sats, _ = block
    s = s0.copy(acqStep=100.0, acqSearchBand=4100.0)
    prns = [sats[0].prn, 5, sats[3].prn]
    a = _run(monkeypatch, s, x, prns)
    b = _run(monkeypatch, s, x, prns, {"BDS_ACQ_PFA": "0"})
    assert (a[1]["rows_kernel"], a[1]["cols_kernel"], a[1]["fft_len"], a[1]["n_bins"]) == (3, 4, N, 83)
    assert b[1]["fft_len"] == 3145728
    _same_decisions(a, b)
    inside = [sat.prn for sat in sats if sat.prn in prns and abs(sat.doppler) < 4000]
    assert inside and all(a[0][0][p - 1] != 0 for p in inside) and a[0][0][4] == 0


def test_peaks_on_the_edges_of_the_three_dimensions(monkeypatch):
    """six satellites whose correlation peaks sit at the first / last index of the 53-, 12- and 3125-point dimensions and on both sides of
    the inter-pass buffer's tile boundaries (lag in the tile 0 / 15, the last tile's 5 lags): lag = t1 N/53 + t2 N/12 + t3 N/3125 mod N"""
    from helpers import spc_of

    s = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1,
                                  acqSearchBand=2100.0)
    spc = spc_of(s)
    edges = [(0, 0, 0), (52, 11, 3124), (26, 5, 3120), (1, 1, 15), (51, 10, 16), (13, 7, 3119)]
    lags = [(t1 * (N // 53) + t2 * (N // 12) + t3 * (N // 3125)) % N for t1, t2, t3 in edges]
    prns = [3, 11, 17, 29, 41, 53]
    # (the sieve's maximum of a synthetic satellite sits two samples behind the sample its code period starts at: the reference's sampled
    #  code tables index with ceil(), B1C/acquisition.m:150-160, and the BOC main lobe is a few samples wide at 99 MS/s)
    sats = [synth.Sat(p, 50.0 * (7 * i - 17), float((t - 2) % spc), 0.3 + i, 50.0) for i, (p, t) in enumerate(zip(prns, lags))]
    x = synth.make_if(s, sats, 4 * spc, seed=41, code_doppler=False)
    a = _run(monkeypatch, s, x, prns + [5])
    b = _run(monkeypatch, s, x, prns + [5], {"BDS_ACQ_PFA": "0"})
    assert (a[1]["rows_kernel"], a[1]["cols_kernel"], a[1]["fft_len"]) == (3, 4, N) and b[1]["fft_len"] == 3145728
    _same_decisions(a, b)
    for i, (p, t) in enumerate(zip(prns, lags)):
        assert a[0][0][p - 1] != 0, p
        bin_ = int(np.argmax(a[2][i]))
        assert int(a[3][i][bin_]) % spc == t % spc == int(b[3][i][bin_]) % spc, (p, edges[i], int(a[3][i][bin_]), t)
        d = (a[0][1][p - 1] - t) % spc  # codePhase (the reference reports a peak at the block's first sample as samplesPerCode)
        assert min(d, spc - d) <= 3.0, (p, a[0][1][p - 1], t % spc)
    assert a[0][0][4] == 0


@pytest.mark.parametrize("change,why", [(dict(acqStep=25.0, acqSearchBand=500.0), "half a spectrum bin per Doppler step"),
                                        (dict(pilotACQflag=0), "one component"),
                                        (dict(acqCohT=5), "N = 15 ms of samples")])
def test_settings_the_n_point_pair_does_not_cover_take_the_l_point_pair(block, monkeypatch, change, why):
    s0, x, sats, _ = block
    s = s0.copy(acqSearchBand=1000.0).copy(**change)
    res, tm, *_ = _run(monkeypatch, s, x, [46, 5])  # PRN 46 is in the block at -308 Hz, PRN 5 is not
    assert tm["rows_kernel"] in (1, 2) and tm["fft_len"] != N, why
    assert res[0][46 - 1] != 0 and res[0][4] == 0


def test_iq_record_at_cfg3_size(monkeypatch):
    """fileType 2 (interleaved I/Q int8, B1C/postProcessing.m:92-96) at 99.375 MS/s: the complex block through the N-point pair"""
    from helpers import as_complex, spc_of

    s = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1, fileType=2,
                                  acqSearchBand=2500.0)
    spc = spc_of(s)
    sats = [synth.Sat(7, -1730.0, 0.613 * spc, 0.7, 45.0), synth.Sat(23, 2210.0, 0.2 * spc, 2.0, 46.0)]
    from bds_amd.acquisition import _as_int8

    x, is_complex = _as_int8(as_complex(synth.make_if(s, sats, 4 * spc, seed=77, iq_sign=-1)), s)  # interleaved I/Q int8, as the file holds it
    assert is_complex
    prns = [7, 8, 23]

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = bds_amd.native.Context(0)
        for k in env:
            monkeypatch.delenv(k)
        try:
            c.acq_load(s, x, True)
            c.acq_prepare(s)
            res = c.acq_run(s, prn_list=prns)
            tm = c.timing()
            grid, _ = c.acq_grid(3, 101)
            pk, _, fb = c.acq_peaks(63)
        finally:
            c.close()
        return res, tm, grid, pk, fb

    a, b = run({}), run({"BDS_ACQ_PFA": "0"})
    assert a[1]["rows_kernel"] == 3 and a[1]["fft_len"] == N and b[1]["fft_len"] == 3145728
    for u, v in zip(a[0], b[0]):
        assert np.array_equal(u, v)
    np.testing.assert_array_equal(a[3], b[3])
    np.testing.assert_array_equal(a[4], b[4])
    np.testing.assert_allclose(a[2], b[2], rtol=2e-3)
    assert a[0][0][6] != 0 and a[0][0][22] != 0 and a[0][0][7] == 0
    assert abs(a[0][0][6] - (s.IF - 1730.0)) <= 25


def test_all_zero_block_falls_back(block, monkeypatch):
    """an all-zero block is one exact tie over every lag: the candidate list runs over and the call is redone on the L-point pair with fp32
    storage, then on the run-time-plan kernels (first-index tie rule): nothing detected, no crash, no stale state for the next call"""
    s0, x, sats, _ = block
    s = s0.copy(acqSearchBand=400.0)
    c = bds_amd.native.Context(0)
    try:
        z = np.zeros(4 * 993750, dtype=np.int8)
        c.acq_load(s, z)
        c.acq_prepare(s)
        res = c.acq_run(s, prn_list=[1, 2])
        assert not np.any(res[0]) and not np.any(res[1])
        c.acq_load(s, x)  # and the same context serves a real block afterwards
        c.acq_prepare(s)
        res = c.acq_run(s, prn_list=[46, 2])
        assert c.timing()["rows_kernel"] == 3  # (the fallback of the zero block did not stick: a new block starts on the N-point pair again)
        assert res[0][46 - 1] != 0 and res[0][1] == 0
    finally:
        c.close()


def test_budget_goes_down_and_the_buffer_with_it(block):
    """bds_acq_set_pair_budget_gb in both directions (round 6): the results stay the same bits, and a context that ran with a larger budget
    gives the memory back at its next run with a smaller one"""
    import torch

    s0, x, sats, _ = block
    s = s0.copy(acqSearchBand=2650.0)  # 107 bins (a "big grid": one PRN per pair at budget 0): 1.73 GB per PRN
    prns = list(range(1, 13))
    c = bds_amd.native.Context(0)
    try:
        c.acq_load(s, x)
        c.acq_prepare(s)
        c.acq_set_pair_budget(12)  # six PRNs per pair: 10.4 GB
        big = c.acq_run(s, prn_list=prns)
        tb = c.timing()
        free_big = torch.cuda.mem_get_info(0)[0]
        c.acq_set_pair_budget(0)
        small = c.acq_run(s, prn_list=prns)
        ts = c.timing()
        free_small = torch.cuda.mem_get_info(0)[0]
    finally:
        c.close()
    for u, v in zip(big, small):
        assert np.array_equal(u, v)
    assert tb["cells_per_pair"] == 6 * 107 and ts["cells_per_pair"] == 107
    assert free_small - free_big > 6 * 2 ** 30, (free_big, free_small)
