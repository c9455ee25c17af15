#!/usr/bin/env python3
"""Extract the literal constant tables the reference's code generators and frame-sync routines
carry -- data the reference itself ships -- into tests/golden/ref_tables.npz, so that the oracle's
re-entered ICD constants and the product's generators are pinned against the reference's own
numbers (a transcription error in either would otherwise be invisible).

    python tests/golden/make_ref_tables.py      (needs /root/reference; build container only)

Tables (all integers):
  b1c_wp_data      [63][2]  (w, p) of the B1C data primary code      generateDataBOC11.m:43-58
  b1c_wp_pilot     [63][2]  ... pilot primary code                   generatePilotBOC11.m:44-59
  b1c_wp_secondary [63][2]  ... pilot secondary code (N = 3607)      generate2ndCode.m:44-58
  b1c_weil_n, b1c_secondary_n                                        generateDataBOC11.m:61, generate2ndCode.m:61
  b2a_reg2_data    [63][13] register-2 initial states (bits)         generateB2aDataCode.m:38-101
  b2a_reg2_pilot   [63][13]                                          generateB2aPilotCode.m:38-101
  b2a_taps_data_r1 / _r2, b2a_taps_pilot_r1 / _r2                    generateB2a{Data,Pilot}Code.m:108-109
  b2a_preamble_bits [24], b2a_second_code [5]                        BCNAV2decoding.m:69-74
"""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/BDS3_B1C_B2a"


def matrix_literal(path, name):
    """`name = [ ... ];` -> 2-D integer array (rows separated by ';', continuation '...')."""
    text = open(os.path.join(REF, path)).read()
    m = re.search(r"^\s*" + re.escape(name) + r"\s*=\s*\[(.*?)\]\s*;", text, re.M | re.S)
    assert m, (path, name)
    body = re.sub(r"\.\.\.[^\n]*", " ", m.group(1))   # continuation (and anything after it on the line)
    body = re.sub(r"%[^\n]*", " ", body)               # comments
    rows = [r.split() for r in body.replace(",", " ").split(";") if r.strip()]
    arr = np.array([[int(t) for t in r] for r in rows], dtype=np.int64)
    return arr


def scalar_literal(path, name):
    text = open(os.path.join(REF, path)).read()
    m = re.search(r"^\s*" + re.escape(name) + r"\s*=\s*(\d+)\s*;", text, re.M)
    assert m, (path, name)
    return int(m.group(1))


out = {
    "b1c_wp_data": matrix_literal("BDS-3_B1C/include/generateDataBOC11.m", "wp_data"),
    "b1c_wp_pilot": matrix_literal("BDS-3_B1C/include/generatePilotBOC11.m", "wp_pilot"),
    "b1c_wp_secondary": matrix_literal("BDS-3_B1C/include/generate2ndCode.m", "wp_pilot"),
    "b1c_weil_n": np.int64(scalar_literal("BDS-3_B1C/include/generateDataBOC11.m", "N")),
    "b1c_secondary_n": np.int64(scalar_literal("BDS-3_B1C/include/generate2ndCode.m", "N")),
    "b2a_reg2_data": matrix_literal("BDS-3_B2a/include/generateB2aDataCode.m", "B2aData_reg2_ini"),
    "b2a_reg2_pilot": matrix_literal("BDS-3_B2a/include/generateB2aPilotCode.m", "B2aData_reg2_ini"),
    "b2a_taps_data_r1": matrix_literal("BDS-3_B2a/include/generateB2aDataCode.m", "reg1_FeedbackPos")[0],
    "b2a_taps_data_r2": matrix_literal("BDS-3_B2a/include/generateB2aDataCode.m", "reg2_FeedbackPos")[0],
    "b2a_taps_pilot_r1": matrix_literal("BDS-3_B2a/include/generateB2aPilotCode.m", "reg1_FeedbackPos")[0],
    "b2a_taps_pilot_r2": matrix_literal("BDS-3_B2a/include/generateB2aPilotCode.m", "reg2_FeedbackPos")[0],
    "b2a_preamble_bits": matrix_literal("BDS-3_B2a/include/BCNAV2decoding.m", "preamble_bits")[0],
    "b2a_second_code": matrix_literal("BDS-3_B2a/include/BCNAV2decoding.m", "secondCode")[0],
}
for k in ("b1c_wp_data", "b1c_wp_pilot", "b1c_wp_secondary"):
    assert out[k].shape == (63, 2), (k, out[k].shape)
for k in ("b2a_reg2_data", "b2a_reg2_pilot"):
    assert out[k].shape == (63, 13) and set(np.unique(out[k])) <= {0, 1}, (k, out[k].shape)
assert out["b2a_preamble_bits"].size == 24 and out["b2a_second_code"].size == 5
np.savez_compressed(os.path.join(HERE, "ref_tables.npz"), **out)
print("wrote ref_tables.npz:", {k: np.asarray(v).shape for k, v in out.items()})
