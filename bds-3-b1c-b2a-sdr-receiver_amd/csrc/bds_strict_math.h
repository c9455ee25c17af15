// Strict carrier arithmetic of the tracking correlator (TrkParams::prec 4 and 5): plain C++ that compiles for the device
// (bds_track.hip) and for the host (tests/test_strict_math.py builds this very header with g++ and checks it
// exhaustively / against libm).  Built with -ffp-contract=off: every operation below rounds once, as written.
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#define BDS_HD __host__ __device__ __forceinline__
#else
#define BDS_HD inline
#endif

namespace bds {
// ---- the strict carrier (PREC 4, the default: per sample; PREC 5 needs it once per lane and pass): sin / cos of the reference's own trigarg(k), without the library calls of PREC 3.
// k / fs correctly rounded from the correctly rounded reciprocal (Markstein: q0 = RN(k y), r = k - q0 fs exactly by FMA,
// q = RN(q0 + r y) = RN(k / fs) when y = RN(1 / fs) and q0 is within an ulp; checked exhaustively on the host for the
// sample counts and rates of the tests, tests/test_abi_and_host.py, and by BDS_DASSERT in the debug build)
BDS_HD double div_by_fs(double k, double fs, double inv_fs) {
    const double q0 = k * inv_fs;
    const double r = fma(-q0, fs, k);
    return fma(r, inv_fs, q0);
}
// sin and cos of |x| < 2^27 to ~1 ulp: n = round(x 2/pi), x - n pi/2 in two FMA steps (the first one is exact: the
// product n P1 cancels against x above 2^-52; n P2 < 2^27 6.1e-17, the third piece of pi/2 is below 2^-80 n), then the
// classic degree-13 / degree-14 minimax polynomials on [-pi/4, pi/4] (coefficients as published with fdlibm's
// k_sin.c / k_cos.c) and the quadrant by the low two bits of n.  Neither a branch nor a table.
BDS_HD void sincos_strict(double x, double &sn, double &cs) {
    const double n = rint(x * 0.63661977236758134308);                                 // 2 / pi
    double r = fma(-n, 1.57079632679489655800e+00, x);                                  // P1 = RN(pi / 2)
    r = fma(-n, 6.12323399573676603587e-17, r);                                         // P2 = RN(pi / 2 - P1)
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);        // S6, S5
    ps = fma(z, ps, 2.75573137070700676789e-06);                                        // S4
    ps = fma(z, ps, -1.98412698298579493134e-04);                                       // S3
    ps = fma(z, ps, 8.33333333332248946124e-03);                                        // S2
    ps = fma(z, ps, -1.66666666666666324348e-01);                                       // S1
    const double s0 = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);        // C6, C5
    pc = fma(z, pc, -2.75573143513906633035e-07);                                       // C4
    pc = fma(z, pc, 2.48015872894767294178e-05);                                        // C3
    pc = fma(z, pc, -1.38888888888741095749e-03);                                       // C2
    pc = fma(z, pc, 4.16666666666666019037e-02);                                        // C1
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c0 = w + (((1.0 - w) - hz) + z * (z * pc));
    const int q = (int)n;
    const double a = (q & 1) ? c0 : s0, b = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

}  // namespace bds
