"""MATLAB built-in semantics the restatement has to mirror.

Test infrastructure (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import numpy as np


def m_round(x: float) -> float:
    """MATLAB round(): half away from zero (NumPy rounds half to even)."""
    return math.floor(abs(x) + 0.5) * (1.0 if x >= 0 else -1.0)


def m_rem(x, y):
    """MATLAB rem(): result has the sign of x (C fmod)."""
    return np.fmod(x, y)


def m_var(x: np.ndarray) -> float:
    """MATLAB var(): normalised by N-1."""
    x = np.asarray(x)
    if np.iscomplexobj(x):  # var of a complex vector: sum |x - mean|^2 / (N-1), real
        return float(np.var(x.astype(np.complex128), ddof=1))
    return float(np.var(x.astype(np.float64), ddof=1))


def m_max_first(x: np.ndarray):
    """[m, i] = max(x): first maximal index, 1-based."""
    i = int(np.argmax(x))
    return x[i], i + 1


def m_atan_div(q: float, i: float) -> float:
    """atan(q / i) with MATLAB's true-division semantics (x/0 -> +-inf, 0/0 -> nan)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.arctan(np.float64(q) / np.float64(i)))
