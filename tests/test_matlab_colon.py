"""MATLAB's colon operator as the oracle restates it (oracle/matlab.py m_colon, after MathWorks' published colonop.m): hand-worked
cases, and the C oracle's per-sample form (oracle/c/trk_oracle.c colon_at) against it on the reference's tcode vectors
(B2a/tracking.m:260-286, B1C/WB_tracking.m:289-317)."""
import ctypes

import numpy as np
import pytest

from oracle import cfast
from oracle.matlab import m_colon, m_colon_parts


def test_known_matlab_elements():
    # 0:0.1:1 -- the textbook case: MATLAB's element 4 is 0.1*3 = 0.30000000000000004 (first half, a + k d), element 7 is
    # 1 - 4*0.1 = 0.6 (second half, from the right end; 6*0.1 = 0.6000000000000001 is NOT what MATLAB holds), the
    # mid-point (n = 10 even) is (0 + 1)/2
    v = m_colon(0.0, 0.1, 1.0)
    assert len(v) == 11
    assert v[3] == 0.1 * 3 == 0.30000000000000004
    assert v[5] == 0.5
    assert v[6] == 1.0 - 4 * 0.1 and v[6] != 6 * 0.1
    assert v[7] == 1.0 - 3 * 0.1 and v[8] == 1.0 - 2 * 0.1 and v[9] == 1.0 - 0.1 and v[10] == 1.0


def test_odd_n_has_no_midpoint_and_halves_are_disjoint():
    # n = 9 intervals (10 elements): 0..4 from the left, 5..9 from the right
    a, d, b = 0.1, 0.1, 1.0
    n, c = m_colon_parts(a, d, b)
    assert n == 9 and c == b
    v = m_colon(a, d, b)
    for k in range(5):
        assert v[k] == a + k * d
        assert v[9 - k] == c - k * d


def test_right_end_is_snapped_only_within_tolerance():
    # a + n d one ulp away from b: snapped (c == b exactly)
    a, d = 0.3, 0.1
    b = np.nextafter(a + 7 * d, 2.0)
    n, c = m_colon_parts(a, d, b)
    assert n == 7 and c == b
    # b well short of a + n d: that element is not produced, and the end is NOT b
    n, c = m_colon_parts(0.0, 0.3, 1.0)
    assert n == 3 and c == 0.0 + 3 * 0.3 and c != 1.0
    # round() up, then the overshoot rule takes the last interval back: (b-a)/d = 2.6 -> round 3 -> 0.9 > 0.8 + tol -> n = 2
    n, c = m_colon_parts(0.0, 0.3, 0.8)
    assert n == 2 and c == 0.0 + 2 * 0.3


def test_empty_and_integer_branches():
    assert len(m_colon(1.0, 0.1, 0.5)) == 0
    np.testing.assert_array_equal(m_colon(0.0, 1.0, 7.0), np.arange(8.0))
    np.testing.assert_array_equal(m_colon(2.0, 3.0, 12.5), np.array([2.0, 5.0, 8.0, 11.0]))


@pytest.mark.parametrize("scale,spc,code_len", [(1.0, 0.5, 10230.0), (2.0, 0.25, 10230.0)], ids=["b2a", "b1c"])
def test_c_per_sample_form_is_the_vector(scale, spc, code_len):
    """the C oracle evaluates element k on the fly; the NumPy oracle builds the whole vector: same doubles, and the diagnostic
    counter agrees with a direct count of the ceil() differences against a + k d"""
    cfast.build()
    L = cfast.lib()
    L.bds_oracle_trk_colon_diff.argtypes = [ctypes.c_long] + [ctypes.c_double] * 4 + [ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double)]
    L.bds_oracle_trk_colon_diff.restype = ctypes.c_int
    rng = np.random.default_rng(5)
    fs = 99.375e6
    for trial in range(40):
        code_freq = (1.023e6 if scale == 2.0 else 10.23e6) * (1 + rng.uniform(-3e-6, 3e-6))
        step = code_freq / fs
        rem = rng.uniform(0, step) if trial else 0.0
        blk = int(np.ceil((code_len - rem) / step))
        counts = (ctypes.c_long * 6)()
        mu = ctypes.c_double()
        assert L.bds_oracle_trk_colon_diff(blk, rem, step, spc, scale, counts, ctypes.byref(mu)) == 0
        kk = np.arange(blk, dtype=np.float64)
        for r, off in enumerate((-spc, 0.0, spc)):
            t = m_colon((rem + off) * scale, step * scale, (((blk - 1) * step + rem) + off) * scale)
            assert len(t) == blk
            plain = (rem + off) * scale + kk * (step * scale)
            assert np.all(np.diff(t) > 0)
            assert counts[r] == int(np.count_nonzero(np.ceil(t) != np.ceil(plain)))
            assert counts[3 + r] == int(np.count_nonzero(np.ceil(t * 6) != np.ceil(plain * 6)))
            # the first half IS a + k d; the second half is within a few ulp of it
            h = (blk - 1) // 2
            np.testing.assert_array_equal(t[:h], plain[:h])
            assert np.max(np.abs(t - plain)) <= 4 * np.spacing(t[-1])
        assert 0 <= mu.value <= 4
