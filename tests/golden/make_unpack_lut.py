#!/usr/bin/env python3
"""Extract the known-answer table of the packed-sample converter from the reference:
B2a/include/unpack_cplx.m holds four literal 256-entry look-up tables (byte -> I1, Q1, I2, Q2).
They are data the reference itself ships, so they pin oracle/unpack.py and the device kernel.

    python tests/golden/make_unpack_lut.py      (needs /root/reference; writes unpack_cplx_lut.npz)
"""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/BDS3_B1C_B2a/BDS-3_B2a/include/unpack_cplx.m"

text = open(SRC).read()
cols = []
for name in ("LUT_I_long1", "LUT_Q_long1", "LUT_I_long2", "LUT_Q_long2"):  # output order 1:4:end .. 4:4:end
    m = re.search(r"^" + name + r" = \[(.*?)\];", text, re.M)
    v = np.array([int(t) for t in m.group(1).split(";")], dtype=np.int8)
    assert v.size == 256
    cols.append(v)
np.savez(os.path.join(HERE, "unpack_cplx_lut.npz"), lut=np.stack(cols, axis=1))  # [256][4]
print("wrote unpack_cplx_lut.npz", np.stack(cols, axis=1)[:4].tolist())
