// Ranging-code generators (host side).
//
// Replaces the interpreted per-chip loops of the reference:
//   BDS-3_B2a/include/generateB2aDataCode.m:104-138, generateB2aPilotCode.m:104-138
//   BDS-3_B1C/include/generateDataBOC11.m:61-91, generatePilotBOC11.m:62-94,
//   generatePilotBOC61.m:62-96 (+ JacobiSymbol.m, which recomputes the same
//   Legendre sequence 10242 times per call -- here it is built once from the
//   set of squares mod 10243).
// The per-PRN tables are BDS ICD constants entered as data.
#include "bds_internal.h"

#include <mutex>
#include <vector>

namespace bds {

// B2a register-2 initial states, 13 bits, stage 1 = MSB.  PRN 1-60 are common to
// the data and pilot generators; 61-63 differ (ICD B2a, table of initial values).
static const uint16_t kB2aReg2Common[60] = {
    0x1025, 0x1034, 0x10AD, 0x114F, 0x1155, 0x11AE, 0x11EE, 0x11FB, 0x1329, 0x13DA,
    0x1435, 0x1444, 0x1455, 0x145B, 0x145C, 0x14A3, 0x14F7, 0x1501, 0x153E, 0x15AB,
    0x15B1, 0x1653, 0x1662, 0x1698, 0x16B6, 0x16F2, 0x16FF, 0x1712, 0x173C, 0x17A1,
    0x17C8, 0x17D4, 0x17EB, 0x17F3, 0x1851, 0x1894, 0x18B7, 0x1911, 0x1919, 0x19AB,
    0x19B1, 0x19D2, 0x1A55, 0x1A74, 0x1ACB, 0x1B57, 0x1C34, 0x1C83, 0x1C8B, 0x1CA3,
    0x1CA8, 0x1D3B, 0x1D97, 0x1E48, 0x1E94, 0x1E99, 0x1EDA, 0x1EF8, 0x1EFF, 0x1FB5};
static const uint16_t kB2aReg2DataTail[3] = {0x0402, 0x1BF5, 0x03D2};
static const uint16_t kB2aReg2PilotTail[3] = {0x1486, 0x05F8, 0x0355};

// B1C Weil code parameters {w, p} per PRN (ICD B1C primary-code tables).
static const uint16_t kB1cWpData[63][2] = {
    {2678, 699},  {4802, 694},  {958, 7318},  {859, 2127},  {3843, 715},  {2232, 6682},
    {124, 7850},  {4352, 5495}, {1816, 1162}, {1126, 7682}, {1860, 6792}, {4800, 9973},
    {2267, 6596}, {424, 2092},  {4192, 19},   {4333, 10151}, {2656, 6297}, {4148, 5766},
    {243, 2359},  {1330, 7136}, {1593, 1706}, {1470, 2128}, {882, 6827},  {3202, 693},
    {5095, 9729}, {2546, 1620}, {1733, 6805}, {4795, 534},  {4577, 712},  {1627, 1929},
    {3638, 5355}, {2553, 6139}, {3646, 6339}, {1087, 1470}, {1843, 6867}, {216, 7851},
    {2245, 1162}, {726, 7659},  {1966, 1156}, {670, 2672},  {4130, 6043}, {53, 2862},
    {4830, 180},  {182, 2663},  {2181, 6940}, {2006, 1645}, {1080, 1582}, {2288, 951},
    {2027, 6878}, {271, 7701},  {915, 1823},  {497, 2391},  {139, 2606},  {3693, 822},
    {2054, 6403}, {4342, 239},  {3342, 442},  {2592, 6769}, {1007, 2560}, {310, 2502},
    {4203, 5072}, {455, 7268},  {4318, 341}};
static const uint16_t kB1cWpPilot[63][2] = {
    {796, 7575},  {156, 2369},  {4198, 5688}, {3941, 539},  {1374, 2270}, {1338, 7306},
    {1833, 6457}, {2521, 6254}, {3175, 5644}, {168, 7119},  {2715, 1402}, {4408, 5557},
    {3160, 5764}, {2796, 1073}, {459, 7001},  {3594, 5910}, {4813, 10060}, {586, 2710},
    {1428, 1546}, {2371, 6887}, {2285, 1883}, {3377, 5613}, {4965, 5062}, {3779, 1038},
    {4547, 10170}, {1646, 6484}, {1430, 1718}, {607, 2535}, {2118, 1158}, {4709, 526},
    {1149, 7331}, {3283, 5844}, {2473, 6423}, {1006, 6968}, {3670, 1280}, {1817, 1838},
    {771, 1989},  {2173, 6468}, {740, 2091},  {1433, 1581}, {2458, 1453}, {3459, 6252},
    {2155, 7122}, {1205, 7711}, {413, 7216},  {874, 2113},  {2463, 1095}, {1106, 1628},
    {1590, 1713}, {3873, 6102}, {4026, 6123}, {4272, 6070}, {3556, 1115}, {128, 8047},
    {1200, 6795}, {130, 2575},  {4494, 53},   {1871, 1729}, {3073, 6388}, {4386, 682},
    {4098, 5565}, {1923, 7160}, {1176, 2277}};

static const int kWeilN = 10243;

// B2a: bit b of a register word = stage (b+1); value 1 <-> chip -1 (the reference
// keeps +-1 and multiplies; here products become XORs of the sign bits).
static void b2a_primary(int prn, bool pilot, int8_t *out) {
    uint16_t ini = prn <= 60 ? kB2aReg2Common[prn - 1]
                             : (pilot ? kB2aReg2PilotTail : kB2aReg2DataTail)[prn - 61];
    // ini has stage 1 in the MSB (bit 12); move stage s to bit s-1.
    uint32_t r2 = 0;
    for (int s = 1; s <= 13; ++s)
        if ((ini >> (13 - s)) & 1) r2 |= 1u << (s - 1);
    uint32_t r1 = 0x1FFF;  // all stages -1
    // tap masks (stage s -> bit s-1): generateB2a*Code.m:108-109
    const uint32_t t1 = pilot ? ((1u << 2) | (1u << 5) | (1u << 6) | (1u << 12))
                              : ((1u << 0) | (1u << 4) | (1u << 10) | (1u << 12));
    const uint32_t t2 =
        pilot ? ((1u << 0) | (1u << 4) | (1u << 6) | (1u << 7) | (1u << 11) | (1u << 12))
              : ((1u << 2) | (1u << 4) | (1u << 8) | (1u << 10) | (1u << 11) | (1u << 12));
    for (int ind = 1; ind <= 10230; ++ind) {
        int bit = ((r1 >> 12) ^ (r2 >> 12)) & 1;  // product of the two stage-13 signs
        out[ind - 1] = bit ? -1 : 1;
        uint32_t f1 = __builtin_parity(r1 & t1);
        uint32_t f2 = __builtin_parity(r2 & t2);
        r1 = ((r1 << 1) | f1) & 0x1FFF;
        r2 = ((r2 << 1) | f2) & 0x1FFF;
        if (ind == 8190) r1 = 0x1FFF;  // generateB2aDataCode.m:135-137
    }
}

// (w, p) of the pilot secondary codes (generate2ndCode.m:44-57), Weil codes over N = 3607
static const uint16_t kB1cWpSecondary[63][2] = {
    {269, 1889}, {1448, 1268}, {1028, 1593}, {1324, 1186}, {822, 1239}, {5, 1930},
    {155, 176}, {458, 1696}, {310, 26}, {959, 1344}, {1238, 1271}, {1180, 1182},
    {1288, 1381}, {334, 1604}, {885, 1333}, {1362, 1185}, {181, 31}, {1648, 704},
    {838, 1190}, {313, 1646}, {750, 1385}, {225, 113}, {1477, 860}, {309, 1656},
    {108, 1921}, {1457, 1173}, {149, 1928}, {322, 57}, {271, 150}, {576, 1214},
    {1103, 1148}, {450, 1458}, {399, 1519}, {241, 1635}, {1045, 1257}, {164, 1687},
    {513, 1382}, {687, 1514}, {422, 1}, {303, 1583}, {324, 1806}, {495, 1664},
    {725, 1338}, {780, 1111}, {367, 1706}, {882, 1543}, {631, 1813}, {37, 228},
    {647, 2871}, {1043, 2884}, {24, 1823}, {120, 75}, {134, 11}, {136, 63},
    {158, 1937}, {214, 22}, {335, 1768}, {340, 1526}, {661, 1402}, {889, 1445},
    {929, 1680}, {1002, 1290}, {1149, 1245}};
static constexpr int kWeilN2 = 3607;

static const std::vector<uint8_t> &legendre() {
    static std::vector<uint8_t> leg;
    static std::once_flag once;
    std::call_once(once, [] {
        leg.assign(kWeilN, 0);
        for (long i = 1; i < kWeilN; ++i) leg[(i * i) % kWeilN] = 1;  // quadratic residues
    });
    return leg;
}

static void b1c_primary(int prn, bool pilot, int8_t *out) {
    const uint16_t *wp = pilot ? kB1cWpPilot[prn - 1] : kB1cWpData[prn - 1];
    const int w = wp[0], p = wp[1];
    const auto &leg = legendre();
    for (int ind = 0; ind < 10230; ++ind) {
        int k = (ind + p - 1) % kWeilN;
        int bit = leg[k] ^ leg[(k + w) % kWeilN];  // generateDataBOC11.m:76-79
        out[ind] = bit ? -1 : 1;
    }
}

// generate2ndCode.m:59-84: 1800 chips, bipolar
int gen_secondary(int prn, int8_t *out) {
    if (prn < 1 || prn > BDS_MAX_PRN) return BDS_ERR_ARG;
    static std::vector<uint8_t> leg;
    static std::once_flag once;
    std::call_once(once, [] {
        leg.assign(kWeilN2, 0);
        for (long i = 1; i < kWeilN2; ++i) leg[(i * i) % kWeilN2] = 1;
    });
    const int w = kB1cWpSecondary[prn - 1][0], p = kB1cWpSecondary[prn - 1][1];
    for (int ind = 0; ind < 1800; ++ind) {
        const int k = (ind + p - 1) % kWeilN2;
        out[ind] = (leg[k] ^ leg[(k + w) % kWeilN2]) ? -1 : 1;
    }
    return 1800;
}

int gen_primary(int signal, bool pilot, int prn, int8_t *out) {
    if (prn < 1 || prn > BDS_MAX_PRN) return BDS_ERR_ARG;
    if (signal == BDS_SIGNAL_B1C)
        b1c_primary(prn, pilot, out);
    else if (signal == BDS_SIGNAL_B2A)
        b2a_primary(prn, pilot, out);
    else
        return BDS_ERR_ARG;
    return 10230;
}

}  // namespace bds

extern "C" int bds_gen_code(int signal, int kind, int prn, int8_t *out, int n) {
    if (!out || prn < 1 || prn > BDS_MAX_PRN) return BDS_ERR_ARG;
    if (kind == BDS_CODE_PILOT_SECONDARY) {
        if (signal != BDS_SIGNAL_B1C || n < 1800) return BDS_ERR_ARG;
        return bds::gen_secondary(prn, out);
    }
    int8_t prim[10230];
    const bool pilot = (kind == BDS_CODE_PILOT_PRIMARY || kind == BDS_CODE_PILOT_BOC11 ||
                        kind == BDS_CODE_PILOT_BOC61);
    int rc = bds::gen_primary(signal, pilot, prn, prim);
    if (rc < 0) return rc;
    switch (kind) {
        case BDS_CODE_DATA_PRIMARY:
        case BDS_CODE_PILOT_PRIMARY:
            if (n < 10230) return BDS_ERR_ARG;
            for (int i = 0; i < 10230; ++i) out[i] = prim[i];
            return 10230;
        case BDS_CODE_DATA_BOC11:
        case BDS_CODE_PILOT_BOC11:  // chip c -> [-c, +c]
            if (signal != BDS_SIGNAL_B1C || n < 20460) return BDS_ERR_ARG;
            for (int i = 0; i < 10230; ++i) {
                out[2 * i] = (int8_t)-prim[i];
                out[2 * i + 1] = prim[i];
            }
            return 20460;
        case BDS_CODE_PILOT_BOC61:  // chip c -> (-1)^ii c, ii = 1..12
            if (signal != BDS_SIGNAL_B1C || n < 122760) return BDS_ERR_ARG;
            for (int i = 0; i < 10230; ++i)
                for (int ii = 1; ii <= 12; ++ii)
                    out[12 * i + ii - 1] = (int8_t)((ii & 1) ? -prim[i] : prim[i]);
            return 122760;
        default:
            return BDS_ERR_ARG;
    }
}
