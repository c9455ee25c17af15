"""Packed 2-bit+2-bit I/Q sample converter (oracle; test infrastructure).

Restates B2a/include/unpack_cplx.m:16-63: every input byte holds two complex samples, the low
nibble first; within a nibble bit 0 is the sign of I, bit 1 the sign of Q, bit 2 the magnitude of I
(1 or 3), bit 3 the magnitude of Q (the generator of the tables, :20-30).  Output: int8
I1, Q1, I2, Q2 per byte -- a fileType-2 record.  PINNED: tests/golden/unpack_cplx_lut.npz holds the
four literal 256-entry tables of the reference file (:32-35), checked in tests/test_unpack.py.
"""
import numpy as np

LUT_I = np.array([1, -1, 1, -1, 3, -3, 3, -3, 1, -1, 1, -1, 3, -3, 3, -3], dtype=np.int8)  # :20
LUT_Q = np.array([1, 1, -1, -1, 1, 1, -1, -1, 3, 3, -3, -3, 3, 3, -3, -3], dtype=np.int8)  # :21


def unpack_cplx(data) -> np.ndarray:
    d = np.asarray(data, dtype=np.uint8).reshape(-1)
    out = np.empty(4 * d.size, dtype=np.int8)
    out[0::4] = LUT_I[d & 15]  # :52
    out[1::4] = LUT_Q[d & 15]
    out[2::4] = LUT_I[d >> 4]
    out[3::4] = LUT_Q[d >> 4]
    return out
