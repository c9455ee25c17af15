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


def m_colon_parts(a: float, d: float, b: float):
    """(n, c) of MATLAB's ``a:d:b`` for non-integer operands: n = number of intervals, c = the right-hand end point.

    Restates the algorithm MathWorks published for the built-in colon operator (``colonop.m``, MathWorks Support
    Technical Solution 1-4FLI96 "How does the COLON operator work?", also reproduced in C. Moler's blog): the vector
    is NOT built by repeated addition and not as ``a + k*d`` throughout --

      tol = 2*eps*max(|a|,|b|)
      n   = round((b-a)/d);  if sign(d)*(a+n*d-b) > tol, n = n-1      (general, non-integer case)
      c   = a + n*d;         if sign(d)*(c-b) > -tol,   c = b         (right end snapped to b)
      out(1+k)   = a + k*d,  k = 0 .. floor(n/2)                       (first half from the left end)
      out(n+1-k) = c - k*d,  k = 0 .. floor(n/2)                       (second half from the RIGHT end)
      if n even: out(n/2+1) = (a+c)/2                                  (mid-point)

    Returns (-1, nan) for an empty result.  Integer operands (the reference's ``0:blksize``, ``1:N``) take colonop's
    integer branches, which give exact consecutive integers: use ``np.arange`` for those.
    """
    a, d, b = float(a), float(d), float(b)
    eps = np.finfo(np.float64).eps
    tol = 2.0 * eps * max(abs(a), abs(b))
    sig = 1.0 if d > 0 else (-1.0 if d < 0 else 0.0)
    if not (math.isfinite(a) and math.isfinite(d) and math.isfinite(b)):
        raise ValueError("colon: non-finite operand")
    if d == 0 or (a < b and d < 0) or (b < a and d > 0):
        return -1, float("nan")
    if a == math.floor(a) and d == 1:
        n = math.floor(b) - a
    elif a == math.floor(a) and d == math.floor(d):
        q = math.floor(a / d)
        r = a - q * d
        n = math.floor((b - r) / d) - q
    else:
        n = m_round((b - a) / d)
        if sig * (a + n * d - b) > tol:
            n = n - 1
    n = int(n)
    c = a + n * d
    if sig * (c - b) > -tol:
        c = b
    return n, c


def m_colon(a: float, d: float, b: float) -> np.ndarray:
    """``a:d:b`` element for element as MATLAB builds it (see m_colon_parts)."""
    n, c = m_colon_parts(a, d, b)
    if n < 0:
        return np.zeros(0)
    a, d = float(a), float(d)
    out = np.empty(n + 1, dtype=np.float64)
    h = n // 2
    k = np.arange(h + 1, dtype=np.float64)
    out[: h + 1] = a + k * d
    out[n - np.arange(h + 1)] = c - k * d
    if n % 2 == 0:
        out[h] = (a + c) / 2
    return out
