// Host check of csrc/bds_strict_math.h (the strict carrier of the tracking correlator): built by tests/test_strict_math.py
// with g++ -O2 -ffp-contract=off -mfma and run on the CPU.
//   1. div_by_fs(k, fs, RN(1/fs)) == k / fs (IEEE division) for EVERY k in [0, kmax) and each rate of the list;
//   2. sincos_strict(x) against libm's sin / cos on trigarg-like arguments: worst error in units of 1e-16.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "bds_strict_math.h"

int main(int argc, char **argv) {
    const long kmax = argc > 1 ? atol(argv[1]) : (1L << 22);
    const double rates[] = {99.375e6, 12.5e6, 25e6, 50e6, 38.192e6, 16.3676e6, 62e6, 102e6, 105e6, 5.456e6, 99.375e6 / 3.0, 1e8 / 7.0};
    long bad = 0;
    for (double fs : rates) {
        const double y = 1.0 / fs;
        for (long k = 0; k < kmax; ++k) {
            const double q = bds::div_by_fs((double)k, fs, y), want = (double)k / fs;
            if (memcmp(&q, &want, 8) != 0) ++bad;
        }
    }
    double worst = 0;
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < 4000000; ++i) {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        const double u = (double)(st >> 11) / 9007199254740992.0;      // [0, 1)
        const double x = (i & 1 ? -1.0 : 1.0) * u * (i % 3 == 0 ? 7.0 : i % 3 == 1 ? 1.0e6 : 6.0e7);
        double s, c;
        bds::sincos_strict(x, s, c);
        const double es = fabs(s - sin(x)), ec = fabs(c - cos(x));
        if (es > worst) worst = es;
        if (ec > worst) worst = ec;
    }
    printf("{\"div_mismatches\": %ld, \"div_checked\": %ld, \"sincos_worst\": %.3e}\n", bad, kmax * (long)(sizeof(rates) / sizeof(rates[0])), worst);
    return 0;
}
