"""The search grid is a sieve: the f64 refinement decides, so the sieve must hand it EVERY cell that could be the
maximum.  bds_acq.hip refines each (bin, lag) whose sieve value is within kDelta of the PRN's sieve maximum; that
is complete when the sieve errs by less than kDelta / 2.  These tests check it directly against the oracle's full
results matrix: every cell within kDelta / 2 of the true maximum must be among the refined candidates
(bds_acq_candidates) -- including cells that share a column tile with a larger neighbour, which the column pass
reports through its overflow list (one record per tile would hide them)."""
import numpy as np
import pytest

import bds_amd
from oracle import acquisition as oacq

from helpers import medium_b2a, small_b1c

pytestmark = pytest.mark.gpu

KDELTA = {0: 2e-5, 1: 4e-3}  # bds_acq.hip, per timing()["half_storage"]


def _oracle_matrix(s, x, prn):
    gen = oacq.b1c_coarse_rows if s.signal.upper() == "B1C" else oacq.b2a_coarse_rows
    return np.stack([row for _, row in gen(x.astype(np.float64), s, prn)])


def _check_complete(ctx, s, x, kdelta, min_cells=1):
    n_found = 0
    for prn in s.acqSatelliteList:
        res = _oracle_matrix(s, x, int(prn))
        m = res.max()
        b, lag = np.nonzero(res >= (1.0 - kdelta / 2) * m)
        want = set(zip((b + 1).tolist(), (lag + 1).tolist()))
        cand = ctx.acq_candidates(int(prn))
        got = set(map(tuple, cand.tolist()))
        missing = want - got
        assert not missing, f"PRN {prn}: {len(missing)} of {len(want)} near-maximum cells were not refined, e.g. {sorted(missing)[:5]}"
        # ... and the global maximum itself, with MATLAB's first-index tie rule
        bb, ll = np.unravel_index(int(np.argmax(res)), res.shape)
        assert (bb + 1, ll + 1) in got
        n_found += len(want)
    assert n_found >= min_cells
    return n_found


@pytest.mark.parametrize("env", [{}, {"BDS_ACQ_FP16": "0"}, {"BDS_ACQ_WCOLS": "1"}, {"BDS_ACQ_WCOLS": "1", "BDS_ACQ_FP16": "0"}])
@pytest.mark.parametrize("case", ["b1c", "b2a"])
def test_every_near_maximum_cell_is_refined(monkeypatch, case, env):
    """(BDS_ACQ_WCOLS=1 forces the wave-private column pass -- per-cell maxima, running bounds and one candidate list instead
    of tile records -- onto these 256-point plans, which default to the tile kernel; at cfg3 it is the default and
    tests/test_fullsize_gpu.py checks it against the whole 201 x 1 987 500 oracle matrix of an absent PRN.)"""
    s, x, _ = small_b1c() if case == "b1c" else medium_b2a()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = bds_amd.native.Context(0)
    try:
        c.acq_load(s, x)
        c.acq_prepare(s)
        c.acq_run(s)
        mode = c.timing()["half_storage"]
        assert mode == (0 if "BDS_ACQ_FP16" in env else 1)
        _check_complete(c, s, x, KDELTA[mode])
    finally:
        c.close()


@pytest.mark.parametrize("wcols", ["0", "1"])
@pytest.mark.parametrize("case", ["b1c", "b2a"])
def test_wide_tolerance_exercises_the_overflow_list(monkeypatch, case, wcols):
    """With the tolerance forced to 20 % thousands of cells pass the sieve (dozens within 10 % of a maximum), many of them in the same column tile as a
    larger one: they can only reach the refinement through the column pass's overflow list.  The results must not
    change (more candidates cannot change an f64 decision)."""
    s, x, _ = small_b1c() if case == "b1c" else medium_b2a()
    base = bds_amd.acquisition(x, s, verbose=False)
    monkeypatch.setenv("BDS_ACQ_KDELTA", "0.2")
    monkeypatch.setenv("BDS_ACQ_WCOLS", wcols)  # tile kernel (records + overflow list) / wave-private kernel (one list)
    c = bds_amd.native.Context(0)
    try:
        c.acq_load(s, x)
        c.acq_prepare(s)
        carr, cph, pm, _ = c.acq_run(s)
        assert c.timing()["n_extra"] > 100
        _check_complete(c, s, x, 0.2, min_cells=30)
    finally:
        c.close()
    np.testing.assert_array_equal(carr, base.carrFreq)
    np.testing.assert_array_equal(cph, base.codePhase)
    np.testing.assert_allclose(pm, base.peakMetric, rtol=1e-12)
