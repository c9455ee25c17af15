"""Frame-synchronisation correlators (oracle; test infrastructure).

Restates B1C/include/BCNAV1decoding.m:66-91 (+ generate2ndCode.m:59-84 through codes.py) and
B2a/include/BCNAV2decoding.m:69-97 with NumPy.  PARITY UNPINNED like the rest of the oracle: the
reference ships no vectors for these either (oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np

from . import codes

B2A_SECOND_CODE = np.array([1, 1, 1, -1, 1], dtype=np.float64)  # BCNAV2decoding.m:69
B2A_PREAMBLE_BITS = np.array([-1, -1, -1, 1, 1, 1, -1, 1, 1, -1, 1, 1, -1, -1, 1, -1, -1, -1, -1, 1, -1, 1, 1, 1],
                             dtype=np.float64)  # :74


def threshold_bits(x) -> np.ndarray:
    """bits(bits > 0) = 1; bits(bits <= 0) = -1  (NaN stays NaN in MATLAB only if never assigned:
    NaN > 0 and NaN <= 0 are both false; tracking never produces NaN in a finished channel)."""
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > 0, 1.0, -1.0)


def xcorr_second_half(a, b) -> np.ndarray:
    """XcorrResult(xcorrLength : 2*xcorrLength-1) of xcorr(a, b): lags 0..M-1, M = max(len)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = max(a.size, b.size)
    ap = np.concatenate([a, np.zeros(m - a.size)])
    bp = np.concatenate([b, np.zeros(m - b.size)])
    full = np.correlate(ap, bp, mode="full")  # lag -(m-1) .. m-1; xcorr(a,b)[lag] = sum a[n+lag] b[n]
    return full[m - 1:]


def b2a_pattern() -> np.ndarray:
    return np.kron(B2A_PREAMBLE_BITS, B2A_SECOND_CODE)  # :78


def frame_sync_b1c(prompt, prn):
    """(XcorrResult, index) -- BCNAV1decoding.m:75-91."""
    r = xcorr_second_half(threshold_bits(prompt), codes.generate_2nd_code(int(prn)))
    return r, np.nonzero(np.abs(r) >= 1799.5)[0] + 1


def frame_sync_b2a(prompt):
    """(tlmXcorrResult second half, index) -- BCNAV2decoding.m:84-97."""
    r = xcorr_second_half(threshold_bits(prompt), b2a_pattern())
    return r, np.nonzero(np.abs(r) > 115)[0] + 1
