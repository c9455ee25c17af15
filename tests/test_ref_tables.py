"""The constant tables the reference's generators carry as literals (tests/golden/ref_tables.npz, extracted
by tests/golden/make_ref_tables.py) against (a) the oracle's re-entered ICD constants and (b) the product's
generators: the codes are re-derived here from the FIXTURE tables with a plain restatement of the
generator loops and compared with bds_gen_code / bds_sync_pattern for every PRN.  CPU only."""
import os

import numpy as np
import pytest

from bds_amd import native
from oracle import codes, framesync

HERE = os.path.dirname(os.path.abspath(__file__))
REF = np.load(os.path.join(HERE, "golden", "ref_tables.npz"))


def test_oracle_tables_equal_the_reference_literals():
    np.testing.assert_array_equal(np.array(codes.B1C_WP_DATA), REF["b1c_wp_data"])      # generateDataBOC11.m:43-58
    np.testing.assert_array_equal(np.array(codes.B1C_WP_PILOT), REF["b1c_wp_pilot"])    # generatePilotBOC11.m:44-59
    np.testing.assert_array_equal(np.array(codes.B1C_WP_SECONDARY), REF["b1c_wp_secondary"])  # generate2ndCode.m:44-58
    assert codes.B1C_WEIL_N == int(REF["b1c_weil_n"]) and codes.B1C_SECONDARY_N == int(REF["b1c_secondary_n"])
    for tab, key in ((codes.B2A_REG2_DATA, "b2a_reg2_data"), (codes.B2A_REG2_PILOT, "b2a_reg2_pilot")):
        bits = np.array([[(v >> (12 - i)) & 1 for i in range(13)] for v in tab])  # first stage = MSB
        np.testing.assert_array_equal(bits, REF[key])                                  # generateB2a*Code.m:38-101
    assert codes.B2A_TAPS["data"] == (tuple(REF["b2a_taps_data_r1"]), tuple(REF["b2a_taps_data_r2"]))     # :108-109
    assert codes.B2A_TAPS["pilot"] == (tuple(REF["b2a_taps_pilot_r1"]), tuple(REF["b2a_taps_pilot_r2"]))
    np.testing.assert_array_equal(framesync.B2A_PREAMBLE_BITS, REF["b2a_preamble_bits"])   # BCNAV2decoding.m:74
    np.testing.assert_array_equal(framesync.B2A_SECOND_CODE, REF["b2a_second_code"])       # :69


# ---- codes re-derived from the fixture (independent of oracle/codes.py) -------------------------------
def _legendre_bits(n):
    """1 where i is a quadratic residue mod prime n (Euler's criterion), index 0 -> 0."""
    return np.array([0] + [1 if pow(i, (n - 1) // 2, n) == 1 else 0 for i in range(1, n)], dtype=np.int64)


def _weil(w, p, n, length, leg):
    """generateDataBOC11.m:69-82: bit(ind) = L(k) xor L(k + w), k = ind + p - 1 (mod N), 1 - 2*bit."""
    k = (np.arange(length) + p - 1) % n
    return 1 - 2 * (leg[k] ^ leg[(k + w) % n])


def _b2a_from_tables(reg2_bits, taps1, taps2, length=10230):
    """generateB2aDataCode.m:112-138 in +-1 arithmetic (XOR = product), register 1 reset after chip 8190."""
    r1 = [-1] * 13
    r2 = [1 - 2 * int(b) for b in reg2_bits]
    out = np.empty(length, dtype=np.int64)
    for i in range(length):
        out[i] = r1[12] * r2[12]
        f1 = int(np.prod([r1[t - 1] for t in taps1]))
        f2 = int(np.prod([r2[t - 1] for t in taps2]))
        r1 = [f1] + r1[:12]
        r2 = [f2] + r2[:12]
        if i + 1 == 8190:
            r1 = [-1] * 13
    return out


@pytest.fixture(scope="module")
def leg_primary():
    return _legendre_bits(int(REF["b1c_weil_n"]))


def test_b1c_primary_codes_from_reference_tables(leg_primary):
    n = int(REF["b1c_weil_n"])
    for prn in range(1, 64):
        for kind, key in (("data", "b1c_wp_data"), ("pilot", "b1c_wp_pilot")):
            w, p = REF[key][prn - 1]
            want = _weil(int(w), int(p), n, 10230, leg_primary)
            np.testing.assert_array_equal(native.gen_code("B1C", kind, prn), want, err_msg=f"{kind} PRN {prn}")


def test_b1c_secondary_codes_from_reference_tables():
    n = int(REF["b1c_secondary_n"])
    leg = _legendre_bits(n)
    for prn in range(1, 64):
        w, p = REF["b1c_wp_secondary"][prn - 1]
        want = _weil(int(w), int(p), n, 1800, leg)
        np.testing.assert_array_equal(native.gen_code("B1C", "pilot_secondary", prn), want, err_msg=f"PRN {prn}")
        np.testing.assert_array_equal(native.sync_pattern("B1C", prn), want)


@pytest.mark.parametrize("prn", [1, 2, 19, 33, 60, 61, 62, 63])
def test_b2a_codes_from_reference_tables(prn):
    d = _b2a_from_tables(REF["b2a_reg2_data"][prn - 1], REF["b2a_taps_data_r1"], REF["b2a_taps_data_r2"])
    p = _b2a_from_tables(REF["b2a_reg2_pilot"][prn - 1], REF["b2a_taps_pilot_r1"], REF["b2a_taps_pilot_r2"])
    np.testing.assert_array_equal(native.gen_code("B2A", "data", prn), d)
    np.testing.assert_array_equal(native.gen_code("B2A", "pilot", prn), p)


def test_b2a_sync_pattern_from_reference_tables():
    want = np.kron(REF["b2a_preamble_bits"], REF["b2a_second_code"])  # BCNAV2decoding.m:78
    np.testing.assert_array_equal(native.sync_pattern("B2A"), want)
