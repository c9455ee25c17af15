"""Ranging codes: structural known-answers for the oracle generators, the committed digests, and
agreement of the product's C++ generators (bds_gen_code, host side) with the oracle for every PRN."""
import json
import os

import numpy as np
import pytest

import bds_amd
from bds_amd import native
from oracle import codes

HERE = os.path.dirname(os.path.abspath(__file__))
S2 = bds_amd.init_settings_b2a()
S1 = bds_amd.init_settings_b1c()


def test_appendix_e_digests():
    """SURVEY.md Appendix E (an independent throw-away restatement made during the survey)."""
    d, p = codes.generate_b2a_data_code(1, S2), codes.generate_b2a_pilot_code(1, S2)
    assert (codes.octal_digest(d[:24]), codes.octal_digest(d[-24:]), d.sum()) == ("26771056", "42646672", 14)
    assert (codes.octal_digest(p[:24]), codes.octal_digest(p[-24:]), p.sum()) == ("26772435", "05133452", -132)
    d, p = codes.b1c_primary(1, "data"), codes.b1c_primary(1, "pilot")
    assert (codes.octal_digest(d[:24]), codes.octal_digest(d[-24:]), d.sum()) == ("53773116", "42711657", 0)
    assert (codes.octal_digest(p[:24]), codes.octal_digest(p[-24:]), p.sum()) == ("71676756", "13053205", 0)


def test_committed_digests_all_prns():
    gold = json.load(open(os.path.join(HERE, "golden", "code_digests.json")))
    for prn in range(1, 64):
        for name, c in (("b2a_data", codes.generate_b2a_data_code(prn, S2)),
                        ("b2a_pilot", codes.generate_b2a_pilot_code(prn, S2)),
                        ("b1c_data", codes.b1c_primary(prn, "data")),
                        ("b1c_pilot", codes.b1c_primary(prn, "pilot"))):
            assert [codes.octal_digest(c[:24]), codes.octal_digest(c[-24:]), int(c.sum())] == gold[f"{name}_{prn}"]


def test_legendre_is_the_jacobi_symbol():
    """generateDataBOC11.m:61-68 fills the sequence with JacobiSymbol(i, 10243)."""
    leg = codes.legendre_sequence()
    assert leg[0] == 0 and leg.sum() == (10243 - 1) // 2
    for i in list(range(1, 400)) + [5121, 10242]:
        assert (codes.jacobi_symbol(i, 10243) == 1) == bool(leg[i])


def test_weil_window_property():
    """Every B1C primary code is a 10230-chip window of the length-10243 Weil sequence of its w."""
    leg = codes.legendre_sequence().astype(np.int64)
    n = 10243
    for prn, kind, tab in ((1, "data", codes.B1C_WP_DATA), (37, "pilot", codes.B1C_WP_PILOT), (63, "data", codes.B1C_WP_DATA)):
        w, p = tab[prn - 1]
        k = np.arange(n)
        weil = 1 - 2 * (leg[k] ^ leg[(k + w) % n])
        np.testing.assert_array_equal(codes.b1c_primary(prn, kind), np.roll(weil, -(p - 1))[:10230])


def test_b2a_register_tables():
    assert codes.B2A_REG2_DATA[:60] == codes.B2A_REG2_PILOT[:60]
    assert all(a != b for a, b in zip(codes.B2A_REG2_DATA[60:], codes.B2A_REG2_PILOT[60:]))
    assert len(codes.B2A_REG2_DATA) == len(codes.B2A_REG2_PILOT) == 63
    assert len(codes.B1C_WP_DATA) == len(codes.B1C_WP_PILOT) == 63


def test_b2a_register1_reset_after_chip_8190():
    """Without the reset (generateB2aDataCode.m:135-137) register 1 would just keep running:
    chips 8191.. equal the product of a restarted register 1 with the running register 2."""
    full = codes._b2a_code(7, "data", 10230)
    other = codes._b2a_code(8, "data", 10230)
    # register 1 is PRN independent: full*other = r2(prn7)*r2(prn8) everywhere, so the product of two
    # PRNs is free of register 1 -- in particular continuous across the reset
    prod = full * other
    assert set(np.unique(prod)) <= {-1, 1}
    # first 13 chips: r1 outputs -1 (all-ones start) so chip = -r2 output = -(1-2*bit13..)
    ini = codes.B2A_REG2_DATA[6]
    assert full[0] == -(1 - 2 * (ini & 1))


def test_boc_expansions():
    prim = codes.b1c_primary(5, "pilot")
    b11 = codes.generate_pilot_boc11(S1, 5)
    b61 = codes.generate_pilot_boc61(S1, 5)
    assert b11.size == 20460 and b61.size == 122760
    np.testing.assert_array_equal(b11[0::2], -prim)
    np.testing.assert_array_equal(b11[1::2], prim)
    np.testing.assert_array_equal(b61.reshape(10230, 12)[:, 0], -prim)
    np.testing.assert_array_equal(b61.reshape(10230, 12)[:, 1], prim)
    np.testing.assert_array_equal(b61.reshape(10230, 12)[:, 11], prim)


def test_sampling_tables():
    s = bds_amd.init_settings_b2a()
    t = codes.make_b2a_data_table(19, s)
    c = codes.generate_b2a_data_code(19, s)
    assert t.size == 99375 and t[0] == c[0] and t[-1] == c[-1]
    s1 = bds_amd.init_settings_b1c(samplingFreq=99.375e6)
    t = codes.make_data_table(s1, 19)
    b = codes.generate_data_boc11(s1, 19)
    assert t.size == 993750 and t[0] == b[0] and t[-1] == b[-1]
    # each half-chip lasts 48 or 49 samples at 99.375 MS/s
    runs = np.diff(np.flatnonzero(np.diff(t) != 0))
    assert runs.min() >= 48 and runs.max() <= 4 * 49


@pytest.mark.parametrize("prn", range(1, 64))
def test_product_codegen_matches_oracle(prn):
    np.testing.assert_array_equal(native.gen_code("B2A", "data", prn), codes.generate_b2a_data_code(prn, S2))
    np.testing.assert_array_equal(native.gen_code("B2A", "pilot", prn), codes.generate_b2a_pilot_code(prn, S2))
    np.testing.assert_array_equal(native.gen_code("B1C", "data", prn), codes.b1c_primary(prn, "data"))
    np.testing.assert_array_equal(native.gen_code("B1C", "pilot", prn), codes.b1c_primary(prn, "pilot"))
    if prn in (1, 30, 63):
        np.testing.assert_array_equal(native.gen_code("B1C", "data_boc11", prn), codes.generate_data_boc11(S1, prn))
        np.testing.assert_array_equal(native.gen_code("B1C", "pilot_boc11", prn), codes.generate_pilot_boc11(S1, prn))
        np.testing.assert_array_equal(native.gen_code("B1C", "pilot_boc61", prn), codes.generate_pilot_boc61(S1, prn))
