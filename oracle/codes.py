"""Ranging-code generators and fs-rate sampling tables (oracle; test infrastructure).

Restates
  B2a/include/generateB2aDataCode.m:37-138, generateB2aPilotCode.m:37-138
  B2a/include/makeB2aDataTable.m:42-67,     makeB2aPilotTable.m:42-67
  B1C/include/generateDataBOC11.m:43-91,    generatePilotBOC11.m:44-94
  B1C/include/generatePilotBOC61.m:44-96,   JacobiSymbol.m:48-125
  B1C/include/makeDataTable.m:45-68,        makePilotTable.m:45-69

The per-PRN constant tables (LFSR register-2 initial states, Weil (w, p)
pairs) are BDS ICD facts, re-entered here as integers.
"""
from __future__ import annotations

import functools

import numpy as np

# --- ICD constants -----------------------------------------------------------
# B2a register-2 initial states, 13 bits, first stage = MSB
# (generateB2aDataCode.m:39-101 / generateB2aPilotCode.m:39-101).  Data and
# pilot share PRN 1-60 and differ for PRN 61-63.
_B2A_REG2_COMMON = [
    0x1025, 0x1034, 0x10AD, 0x114F, 0x1155, 0x11AE, 0x11EE, 0x11FB, 0x1329, 0x13DA,
    0x1435, 0x1444, 0x1455, 0x145B, 0x145C, 0x14A3, 0x14F7, 0x1501, 0x153E, 0x15AB,
    0x15B1, 0x1653, 0x1662, 0x1698, 0x16B6, 0x16F2, 0x16FF, 0x1712, 0x173C, 0x17A1,
    0x17C8, 0x17D4, 0x17EB, 0x17F3, 0x1851, 0x1894, 0x18B7, 0x1911, 0x1919, 0x19AB,
    0x19B1, 0x19D2, 0x1A55, 0x1A74, 0x1ACB, 0x1B57, 0x1C34, 0x1C83, 0x1C8B, 0x1CA3,
    0x1CA8, 0x1D3B, 0x1D97, 0x1E48, 0x1E94, 0x1E99, 0x1EDA, 0x1EF8, 0x1EFF, 0x1FB5,
]
B2A_REG2_DATA = _B2A_REG2_COMMON + [0x0402, 0x1BF5, 0x03D2]
B2A_REG2_PILOT = _B2A_REG2_COMMON + [0x1486, 0x05F8, 0x0355]
# feedback tap positions, 1-based stage numbers (generateB2a*Code.m:108-109)
B2A_TAPS = {
    "data": ((1, 5, 11, 13), (3, 5, 9, 11, 12, 13)),
    "pilot": ((3, 6, 7, 13), (1, 5, 7, 8, 12, 13)),
}

# B1C Weil-code (w, p) per PRN (generateDataBOC11.m:43-58, generatePilotBOC11.m:44-59)
B1C_WP_DATA = [
    (2678, 699), (4802, 694), (958, 7318), (859, 2127), (3843, 715), (2232, 6682),
    (124, 7850), (4352, 5495), (1816, 1162), (1126, 7682), (1860, 6792), (4800, 9973),
    (2267, 6596), (424, 2092), (4192, 19), (4333, 10151), (2656, 6297), (4148, 5766),
    (243, 2359), (1330, 7136), (1593, 1706), (1470, 2128), (882, 6827), (3202, 693),
    (5095, 9729), (2546, 1620), (1733, 6805), (4795, 534), (4577, 712), (1627, 1929),
    (3638, 5355), (2553, 6139), (3646, 6339), (1087, 1470), (1843, 6867), (216, 7851),
    (2245, 1162), (726, 7659), (1966, 1156), (670, 2672), (4130, 6043), (53, 2862),
    (4830, 180), (182, 2663), (2181, 6940), (2006, 1645), (1080, 1582), (2288, 951),
    (2027, 6878), (271, 7701), (915, 1823), (497, 2391), (139, 2606), (3693, 822),
    (2054, 6403), (4342, 239), (3342, 442), (2592, 6769), (1007, 2560), (310, 2502),
    (4203, 5072), (455, 7268), (4318, 341),
]
B1C_WP_PILOT = [
    (796, 7575), (156, 2369), (4198, 5688), (3941, 539), (1374, 2270), (1338, 7306),
    (1833, 6457), (2521, 6254), (3175, 5644), (168, 7119), (2715, 1402), (4408, 5557),
    (3160, 5764), (2796, 1073), (459, 7001), (3594, 5910), (4813, 10060), (586, 2710),
    (1428, 1546), (2371, 6887), (2285, 1883), (3377, 5613), (4965, 5062), (3779, 1038),
    (4547, 10170), (1646, 6484), (1430, 1718), (607, 2535), (2118, 1158), (4709, 526),
    (1149, 7331), (3283, 5844), (2473, 6423), (1006, 6968), (3670, 1280), (1817, 1838),
    (771, 1989), (2173, 6468), (740, 2091), (1433, 1581), (2458, 1453), (3459, 6252),
    (2155, 7122), (1205, 7711), (413, 7216), (874, 2113), (2463, 1095), (1106, 1628),
    (1590, 1713), (3873, 6102), (4026, 6123), (4272, 6070), (3556, 1115), (128, 8047),
    (1200, 6795), (130, 2575), (4494, 53), (1871, 1729), (3073, 6388), (4386, 682),
    (4098, 5565), (1923, 7160), (1176, 2277),
]
B1C_WEIL_N = 10243


# --- B2a ----------------------------------------------------------------------
def _b2a_code(prn: int, kind: str, code_length: int = 10230) -> np.ndarray:
    """generateB2aDataCode.m:104-138 / generateB2aPilotCode.m:104-138.

    Two 13-stage shift registers in +-1 arithmetic; chip = r1(13)*r2(13);
    feedback = product of the tapped stages, shifted in at stage 1; register 1
    is reset to all -1 right after chip 8190 has been produced and shifted.
    """
    ini = (B2A_REG2_DATA if kind == "data" else B2A_REG2_PILOT)[prn - 1]
    taps1, taps2 = B2A_TAPS[kind]
    reg1 = [-1] * 13
    reg2 = [1 - 2 * ((ini >> (12 - i)) & 1) for i in range(13)]
    code = np.zeros(code_length, dtype=np.int8)
    for ind in range(1, code_length + 1):
        code[ind - 1] = reg1[12] * reg2[12]
        f1 = 1
        for t in taps1:
            f1 *= reg1[t - 1]
        reg1 = [f1] + reg1[:12]
        f2 = 1
        for t in taps2:
            f2 *= reg2[t - 1]
        reg2 = [f2] + reg2[:12]
        if ind == 8190:
            reg1 = [-1] * 13
    return code


@functools.lru_cache(maxsize=None)
def _b2a_code_cached(prn: int, kind: str, code_length: int):
    c = _b2a_code(prn, kind, code_length)
    c.setflags(write=False)
    return c


def generate_b2a_data_code(prn: int, settings) -> np.ndarray:
    """B2a/include/generateB2aDataCode.m:1 -> 1 x codeLength array of +-1."""
    return _b2a_code_cached(int(prn), "data", int(settings.codeLength)).astype(np.float64)


def generate_b2a_pilot_code(prn: int, settings) -> np.ndarray:
    """B2a/include/generateB2aPilotCode.m:1 -> 1 x codeLength array of +-1."""
    return _b2a_code_cached(int(prn), "pilot", int(settings.codeLength)).astype(np.float64)


def samples_per_code(settings) -> int:
    """round(fs / (codeFreqBasis / codeLength)) (B2a/acquisition.m:130-131)."""
    from .matlab import m_round

    return int(m_round(settings.samplingFreq / (settings.codeFreqBasis / settings.codeLength)))


def _b2a_table(code: np.ndarray, settings) -> np.ndarray:
    """makeB2aDataTable.m:42-67: idx = ceil(ts*(1:spc)/tc); idx(end) = codeLength."""
    spc = samples_per_code(settings)
    ts = 1.0 / settings.samplingFreq
    tc = 1.0 / settings.codeFreqBasis
    idx = np.ceil((ts * np.arange(1, spc + 1, dtype=np.float64)) / tc).astype(np.int64)
    idx[-1] = int(settings.codeLength)
    return code[idx - 1]


def make_b2a_data_table(prn: int, settings) -> np.ndarray:
    return _b2a_table(generate_b2a_data_code(prn, settings), settings)


def make_b2a_pilot_table(prn: int, settings) -> np.ndarray:
    return _b2a_table(generate_b2a_pilot_code(prn, settings), settings)


# (w, p) of the B1C pilot secondary codes, PRN 1..63 (generate2ndCode.m:44-57; BDS-SIS-ICD-B1C table)
B1C_WP_SECONDARY = [
    (269, 1889), (1448, 1268), (1028, 1593), (1324, 1186), (822, 1239), (5, 1930),
    (155, 176), (458, 1696), (310, 26), (959, 1344), (1238, 1271), (1180, 1182),
    (1288, 1381), (334, 1604), (885, 1333), (1362, 1185), (181, 31), (1648, 704),
    (838, 1190), (313, 1646), (750, 1385), (225, 113), (1477, 860), (309, 1656),
    (108, 1921), (1457, 1173), (149, 1928), (322, 57), (271, 150), (576, 1214),
    (1103, 1148), (450, 1458), (399, 1519), (241, 1635), (1045, 1257), (164, 1687),
    (513, 1382), (687, 1514), (422, 1), (303, 1583), (324, 1806), (495, 1664),
    (725, 1338), (780, 1111), (367, 1706), (882, 1543), (631, 1813), (37, 228),
    (647, 2871), (1043, 2884), (24, 1823), (120, 75), (134, 11), (136, 63),
    (158, 1937), (214, 22), (335, 1768), (340, 1526), (661, 1402), (889, 1445),
    (929, 1680), (1002, 1290), (1149, 1245),
]
B1C_SECONDARY_N = 3607  # generate2ndCode.m:61


# --- B1C ----------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def legendre_sequence(n: int = B1C_WEIL_N) -> np.ndarray:
    """legendre(i+1) = 1 if i is a quadratic residue mod N (i != 0) else 0.

    generateDataBOC11.m:61-68 fills it with JacobiSymbol(ind, N) and maps -1 -> 0;
    for prime N the Jacobi symbol is the Legendre symbol, so the set of squares
    mod N gives the same sequence.
    """
    leg = np.zeros(n, dtype=np.int8)
    i = np.arange(1, n, dtype=np.int64)
    leg[(i * i) % n] = 1
    leg.setflags(write=False)
    return leg


def jacobi_symbol(a: int, b: int) -> int:
    """Binary Jacobi algorithm -- only used by tests to cross-check
    legendre_sequence() against the symbol JacobiSymbol.m:48-125 computes."""
    assert b > 0 and b % 2 == 1
    a %= b
    result = 1
    while a != 0:
        while a % 2 == 0:
            a //= 2
            if b % 8 in (3, 5):
                result = -result
        a, b = b, a
        if a % 4 == 3 and b % 4 == 3:
            result = -result
        a %= b
    return result if b == 1 else 0


def b1c_primary(prn: int, kind: str, code_length: int = 10230) -> np.ndarray:
    """generateDataBOC11.m:69-82 / generatePilotBOC11.m:72-85: bipolar primary code.

    chip(ind) = L[k] xor L[(k+w) mod N], k = (ind + p - 1) mod N, ind = 0..10229;
    then 1 - 2*bit.
    """
    w, p = (B1C_WP_DATA if kind == "data" else B1C_WP_PILOT)[prn - 1]
    leg = legendre_sequence()
    n = B1C_WEIL_N
    ind = np.arange(code_length, dtype=np.int64)
    k = (ind + p - 1) % n
    bits = leg[k] ^ leg[(k + w) % n]
    return (1 - 2 * bits.astype(np.int64)).astype(np.float64)


def generate_2nd_code(prn: int) -> np.ndarray:
    """generate2ndCode.m:59-84: 1800-chip pilot secondary code, bipolar.

    bit(ind) = L[k] xor L[(k+w) mod N], k = (ind + p - 1) mod N, N = 3607, ind = 0..1799."""
    w, p = B1C_WP_SECONDARY[prn - 1]
    n = B1C_SECONDARY_N
    leg = legendre_sequence(n)
    ind = np.arange(1800, dtype=np.int64)
    k = (ind + p - 1) % n
    bits = leg[k] ^ leg[(k + w) % n]
    return (1 - 2 * bits.astype(np.int64)).astype(np.float64)


def _boc11(primary: np.ndarray) -> np.ndarray:
    """generateDataBOC11.m:85-91: every chip c -> [-c, +c]."""
    out = np.empty(primary.size * 2, dtype=np.float64)
    out[0::2] = -primary
    out[1::2] = primary
    return out


def generate_data_boc11(settings, prn: int) -> np.ndarray:
    """B1C/include/generateDataBOC11.m:1 -> 1 x 2*codeLength (+-1)."""
    return _boc11(b1c_primary(int(prn), "data", int(settings.codeLength)))


def generate_pilot_boc11(settings, prn: int) -> np.ndarray:
    """B1C/include/generatePilotBOC11.m:1 -> 1 x 2*codeLength (+-1)."""
    return _boc11(b1c_primary(int(prn), "pilot", int(settings.codeLength)))


def generate_pilot_boc61(settings, prn: int) -> np.ndarray:
    """B1C/include/generatePilotBOC61.m:89-96: chip c -> (-1)^ii * c, ii = 1..12."""
    primary = b1c_primary(int(prn), "pilot", int(settings.codeLength))
    sub = np.array([(-1.0) ** ii for ii in range(1, 13)])
    return (primary[:, None] * sub[None, :]).reshape(-1)


def _b1c_table(code: np.ndarray, settings) -> np.ndarray:
    """makeDataTable.m:45-68: tc = 1/codeFreqBasis/2; idx = ceil(ts*(1:spc)/tc);
    idx(end) = 2*codeLength; idx(1) = 1."""
    spc = samples_per_code(settings)
    ts = 1.0 / settings.samplingFreq
    tc = 1.0 / settings.codeFreqBasis / 2
    idx = np.ceil((ts * np.arange(1, spc + 1, dtype=np.float64)) / tc).astype(np.int64)
    idx[-1] = int(settings.codeLength) * 2
    idx[0] = 1
    return code[idx - 1]


def make_data_table(settings, prn: int) -> np.ndarray:
    return _b1c_table(generate_data_boc11(settings, prn), settings)


def make_pilot_table(settings, prn: int) -> np.ndarray:
    return _b1c_table(generate_pilot_boc11(settings, prn), settings)


def octal_digest(chips: np.ndarray) -> str:
    """24 chips -> 8 octal digits, bit = (1-chip)/2, first chip = MSB (the
    commented-out check block at generatePilotBOC61.m:98-106)."""
    bits = ((1 - np.asarray(chips, dtype=np.int64)) // 2).tolist()
    return "".join(str(bits[i] * 4 + bits[i + 1] * 2 + bits[i + 2]) for i in range(0, 24, 3))
