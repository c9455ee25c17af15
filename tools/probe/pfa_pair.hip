// Probe of the N-point search pair (csrc/bds_acq_pfa.h): correctness of the row pass (k_pfa_rows) and of the column
// pass (k_pfa_cols: 53-point stage on the matrix pipe) against direct float64 evaluation, then their time per 201 cells at cfg3 sizes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -I bds-3-b1c-b2a-sdr-receiver_amd/csrc
//         tools/probe/pfa_pair.hip -o gpurun_out/pfa_pair && gpurun_out/pfa_pair [prns] [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "bds_acq_pfa.h"

using namespace bds::pfa;
typedef std::complex<double> cd;

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
            exit(2);                                                                                  \
        }                                                                                             \
    } while (0)

static uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }
static float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
static uint32_t pack(float re, float im) { return (uint32_t)f2h(re) | ((uint32_t)f2h(im) << 16); }
static cd unpack(uint32_t u) { return cd(h2f((uint16_t)(u & 0xffff)), h2f((uint16_t)(u >> 16))); }

struct ArrayLoader {  // the forward kernels' input: x[n] of transform `batch`
    const float2 *x;
    __device__ __forceinline__ float2 operator()(int batch, long n) const { return x[(size_t)batch * NP + n]; }
};

static void check_forward() {
    std::mt19937_64 rng(11);
    std::normal_distribution<float> nd(0.f, 1.f);
    const int nb = 2;
    std::vector<float2> x((size_t)nb * NP);
    for (auto &v : x) v = make_float2(std::round(20.f * nd(rng)), std::round(3.f * nd(rng)));
    float2 *d_x, *d_tmp;
    uint32_t *d_sig, *d_code;
    CK(hipMalloc(&d_x, x.size() * sizeof(float2)));
    CK(hipMalloc(&d_tmp, (size_t)2 * nb * NP * sizeof(float2)));
    CK(hipMalloc(&d_sig, (size_t)2 * NP * 4));
    CK(hipMalloc(&d_code, (size_t)nb * NP * 4));
    CK(hipMemcpy(d_x, x.data(), x.size() * sizeof(float2), hipMemcpyHostToDevice));
    const float sc = 1.0f / 65536.f;
    forward(0, ArrayLoader{d_x}, 1, d_tmp, d_sig, 0, 0, sc, 1);
    forward(0, ArrayLoader{d_x}, nb, d_tmp, d_code, NP, 1, sc, 0);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> sig((size_t)2 * NP), code((size_t)nb * NP);
    CK(hipMemcpy(sig.data(), d_sig, sig.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(code.data(), d_code, code.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, rms = 0;
    int n = 0;
    std::uniform_int_distribution<long> uk(0, NP - 1);
    for (int trial = 0; trial < 12; ++trial) {
        const long k = trial == 0 ? 0 : trial == 1 ? NP - 1 : uk(rng);
        const int k1 = k % K1, k2 = k % K2, k3 = k % K3;
        for (int b = 0; b < nb; ++b) {
            cd ref = 0;
            for (long m = 0; m < NP; ++m) {
                const __int128 ph = ((__int128)m * k) % NP;
                ref += cd(x[(size_t)b * NP + m].x, x[(size_t)b * NP + m].y) * std::polar(1.0, -2 * M_PI * (double)(long)ph / NP);
            }
            ref *= sc;
            const cd gc = unpack(code[(size_t)b * NP + ((size_t)k1 * K2 + k2) * K3 + k3]);
            worst = std::max(worst, std::abs(gc - std::conj(ref)));
            if (b == 0) {
                const cd g0 = unpack(sig[((size_t)k1 * K2 + k2) * 2 * K3 + k3]), g1 = unpack(sig[((size_t)k1 * K2 + k2) * 2 * K3 + K3 + k3]);
                worst = std::max(worst, std::max(std::abs(g0 - ref), std::abs(g1 - ref)));
            }
            rms += std::norm(ref), ++n;
        }
    }
    printf("forward transforms (CRT layout, fp16 storage): worst |error| %.3g against an rms value of %.3g (%.2e)\n", worst, std::sqrt(rms / n), worst / std::sqrt(rms / n));
    CK(hipFree(d_x));
    CK(hipFree(d_tmp));
    CK(hipFree(d_sig));
    CK(hipFree(d_code));
}

int main(int argc, char **argv) {
    check_forward();
    const int prns = argc > 1 ? atoi(argv[1]) : 4, reps = argc > 2 ? atoi(argv[2]) : 3, D = 201;
    const int nslots = std::max(prns, 2);
    std::mt19937_64 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    // spectra in natural order (fp16-exact values), then the CRT layouts the kernels read
    std::vector<uint32_t> Xnat(NP), Cnat((size_t)nslots * 2 * NP);
    for (long k = 0; k < NP; ++k) Xnat[k] = pack(8.f * nd(rng), 8.f * nd(rng));
    for (size_t i = 0; i < Cnat.size(); ++i) Cnat[i] = pack(0.125f * nd(rng), 0.125f * nd(rng));
    std::vector<uint32_t> Xs((size_t)K1 * K2 * 2 * K3), Cs(Cnat.size());
    for (long k = 0; k < NP; ++k) {
        const int k1 = k % K1, k2 = k % K2, k3 = k % K3;
        Xs[((size_t)k1 * K2 + k2) * 2 * K3 + k3] = Xnat[k];
        Xs[((size_t)k1 * K2 + k2) * 2 * K3 + K3 + k3] = Xnat[k];
        for (int sc = 0; sc < nslots * 2; ++sc) Cs[(size_t)sc * NP + ((size_t)k1 * K2 + k2) * K3 + k3] = Cnat[(size_t)sc * NP + k];
    }
    // cells of the correctness run
    const int tb[6] = {0, 1, 5, 57, 100, 200}, tslot[6] = {0, 0, 0, 1, 1, 1};
    const int ncell_t = 6;
    const int ncells = prns * D;
    std::vector<int> h_bin(std::max(ncells, ncell_t));
    std::vector<long> h_cs(std::max(ncells, ncell_t));
    uint32_t *d_Xs, *d_Cs, *d_Bw;
    int *d_bin;
    long *d_cs;
    uint4 *d_coef;
    unsigned long long *d_cellmax, *d_stats;
    float *d_lb, *d_dbg;
    bds::Extra *d_extra;
    int *d_extra_count;
    const int extra_cap = 1 << 22;
    CK(hipMalloc(&d_Xs, Xs.size() * 4));
    CK(hipMalloc(&d_Cs, Cs.size() * 4));
    CK(hipMalloc(&d_Bw, (size_t)std::max(ncells, ncell_t) * kCellElems * 4));
    CK(hipMalloc(&d_bin, h_bin.size() * sizeof(int)));
    CK(hipMalloc(&d_cs, h_cs.size() * sizeof(long)));
    CK(hipMalloc(&d_coef, kCoefBytes));
    CK(hipMalloc(&d_cellmax, h_bin.size() * sizeof(unsigned long long)));
    CK(hipMalloc(&d_lb, h_bin.size() * sizeof(float)));
    CK(hipMalloc(&d_stats, 4 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_extra, sizeof(bds::Extra) * extra_cap));
    CK(hipMalloc(&d_extra_count, sizeof(int)));
    CK(hipMalloc(&d_dbg, sizeof(float) * 2 * K1 * 12 * 4));
    CK(hipMemcpy(d_Xs, Xs.data(), Xs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_Cs, Cs.data(), Cs.size() * 4, hipMemcpyHostToDevice));
    {
        std::vector<uint16_t> cf(kCoefBytes / 2);
        make_coef_frags(cf.data());
        CK(hipMemcpy(d_coef, cf.data(), kCoefBytes, hipMemcpyHostToDevice));
    }
    const size_t rows_lds = 2 * 3136 * sizeof(float2);
    CK(hipFuncSetAttribute((const void *)k_pfa_cols<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColsLds));
    CK(hipFuncSetAttribute((const void *)k_pfa_cols<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColsLds));

    // ---------------- correctness: each test cell is its own one-cell chunk (gc = 1: the PRN slot changes from cell to cell)
    for (int i = 0; i < ncell_t; ++i) h_bin[i] = tb[i], h_cs[i] = (long)tslot[i] * 2 * NP;
    CK(hipMemcpy(d_bin, h_bin.data(), ncell_t * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cs, h_cs.data(), ncell_t * sizeof(long), hipMemcpyHostToDevice));
    CK(hipMemset(d_Bw, 0xff, (size_t)ncell_t * kCellElems * 4));  // NaN pattern: every element must be written
    RowsArgs ra{d_Xs, d_Cs, d_Bw, d_bin, d_cs, ncell_t, 1, 1};
    hipLaunchKernelGGL(k_pfa_rows<2>, dim3(MP * K2 * ncell_t), dim3(kRowsThreads), rows_lds, 0, ra);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> Bw((size_t)ncell_t * kCellElems);
    CK(hipMemcpy(Bw.data(), d_Bw, Bw.size() * 4, hipMemcpyDeviceToHost));
    auto Ycrt = [&](int cell, int comp, int k1, int k2, int k3) {  // product spectrum of a cell at CRT coordinates
        const int s = tb[cell];
        const int a = ((k1 - s) % K1 + K1) % K1, b = ((k2 - s) % K2 + K2) % K2, c = ((k3 - s) % K3 + K3) % K3;
        return unpack(Xs[((size_t)a * K2 + b) * 2 * K3 + c]) * unpack(Cs[((size_t)tslot[cell] * 2 + comp) * NP + ((size_t)k1 * K2 + k2) * K3 + k3]);
    };
    double worst_rows = 0, rms_rows = 0;
    long nchk = 0;
    std::uniform_int_distribution<int> u1(0, K1), u2(0, K2 - 1), u3(0, K3 - 1);
    for (int cell = 0; cell < ncell_t; ++cell)
        for (int trial = 0; trial < 40; ++trial) {
            const int k1 = trial == 0 ? 53 : trial == 1 ? 52 : u1(rng) % 54, k2 = u2(rng), t3 = trial < 4 ? (trial & 1 ? K3 - 1 : 0) : u3(rng);
            for (int comp = 0; comp < 2; ++comp) {
                cd ref = 0;
                if (k1 < K1)
                    for (int k3 = 0; k3 < K3; ++k3) ref += Ycrt(cell, comp, k1, k2, k3) * std::polar(1.0, 2 * M_PI * (double)(((long)k3 * t3) % K3) / K3);
                const cd got = unpack(Bw[(size_t)cell * kCellElems + bw_piece(k1 / 2, k2, t3) + comp * 2 + (k1 & 1)]);
                worst_rows = std::max(worst_rows, std::abs(got - ref));
                rms_rows += std::norm(ref);
                ++nchk;
            }
        }
    rms_rows = std::sqrt(rms_rows / nchk);
    printf("row pass: %ld sampled outputs of %d cells: worst |error| %.3g against an rms output of %.3g (%.2e; fp16 rounding is 4.9e-4 of a value)\n", nchk, ncell_t,
           worst_rows, rms_rows, worst_rows / rms_rows);
    size_t nan_left = 0;
    for (int cell = 0; cell < ncell_t; ++cell)
        for (int mp = 0; mp < MP; ++mp)
            for (int k2 = 0; k2 < K2; ++k2)
                for (int t3 = 0; t3 < K3; ++t3)
                    for (int e = 0; e < 4; ++e) nan_left += Bw[(size_t)cell * kCellElems + bw_piece(mp, k2, t3) + e] == 0xffffffffu;
    printf("          elements of the inter-pass buffer never written: %zu of %zu\n", nan_left, (size_t)ncell_t * MP * K2 * K3 * 4);

    // column pass on the GPU's own buffer, one (cell, t3 group) at a time in debug mode
    double worst_cols = 0, worst_e2e = 0, big = 0;
    for (int cell : {0, 3, 5})
        for (int grp : {0, 311, 781}) {
            CK(hipMemset(d_cellmax, 0, ncell_t * sizeof(unsigned long long)));
            CK(hipMemset(d_lb, 0, ncell_t * sizeof(float)));
            CK(hipMemset(d_extra_count, 0, sizeof(int)));
            CK(hipMemset(d_dbg, 0, sizeof(float) * 2 * K1 * 12 * 4));
            ColsArgs ca{d_Bw, d_coef, ncell_t, 0.52440442f, 0.85146932f, d_cellmax, d_lb, 1, d_extra, d_extra_count, extra_cap, 0, 0.996f, 4, nullptr, d_dbg, cell, grp};
            hipLaunchKernelGGL((k_pfa_cols<2, true>), dim3(512), dim3(kColsThreads), kColsLds, 0, ca);
            CK(hipDeviceSynchronize());
            std::vector<float> dbg(2 * K1 * 12 * 4);
            CK(hipMemcpy(dbg.data(), d_dbg, dbg.size() * 4, hipMemcpyDeviceToHost));
            for (int comp = 0; comp < 2; ++comp)
                for (int g = 0; g < 4; ++g) {
                    const int t3 = 4 * grp + g;
                    if (t3 >= K3) continue;
                    for (int t1 : {0, 1, 17, 52})
                        for (int t2 : {0, 1, 5, 6, 7, 11}) {
                            cd y = 0;
                            for (int k1 = 0; k1 < K1; ++k1)
                                for (int k2 = 0; k2 < K2; ++k2)
                                    y += unpack(Bw[(size_t)cell * kCellElems + bw_piece(k1 / 2, k2, t3) + comp * 2 + (k1 & 1)]) *
                                         std::polar(1.0, 2 * M_PI * ((double)((k1 * t1) % K1) / K1 + (double)((k2 * t2) % K2) / K2));
                            const double got = dbg[((comp * K1 + t1) * 12 + t2) * 4 + g];
                            worst_cols = std::max(worst_cols, std::abs(got - std::norm(y)));
                            big = std::max(big, std::norm(y));
                        }
                }
            // end to end for two lags of this group: y[t] = sum_k Y[k] exp(+2 pi j k t / N) in natural order
            for (int pick = 0; pick < 2; ++pick) {
                const int t1 = pick ? 52 : 17, t2 = pick ? 11 : 5, g = pick ? 0 : 1, t3 = 4 * grp + g;
                if (t3 >= K3) continue;
                const long t = lag_of(t1, t2, t3);
                const int s = tb[cell];
                for (int comp = 0; comp < 2; ++comp) {
                    cd y = 0;
                    for (long k = 0; k < NP; ++k) {
                        const long ks = ((k - s) % NP + NP) % NP;
                        const __int128 ph = ((__int128)k * t) % NP;
                        y += unpack(Xnat[ks]) * unpack(Cnat[((size_t)tslot[cell] * 2 + comp) * NP + k]) * std::polar(1.0, 2 * M_PI * (double)(long)ph / NP);
                    }
                    const double got = dbg[((comp * K1 + t1) * 12 + t2) * 4 + g];
                    worst_e2e = std::max(worst_e2e, std::abs(std::sqrt(got) - std::abs(y)) / std::abs(y));
                }
            }
        }
    {   // the sieve's outputs of the last of those launches against a float64 evaluation of two whole cells from the same buffer
        std::vector<unsigned long long> cm(ncell_t);
        int n_ex = 0;
        CK(hipMemcpy(cm.data(), d_cellmax, ncell_t * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&n_ex, d_extra_count, sizeof(int), hipMemcpyDeviceToHost));
        std::vector<bds::Extra> ex(std::min(n_ex, extra_cap));
        CK(hipMemcpy(ex.data(), d_extra, ex.size() * sizeof(bds::Extra), hipMemcpyDeviceToHost));
        const double w0 = 0.52440442, w1 = 0.85146932, keep = 0.996;
        std::vector<cd> W53(K1), W12(K2);
        for (int i = 0; i < K1; ++i) W53[i] = std::polar(1.0, 2 * M_PI * i / K1);
        for (int i = 0; i < K2; ++i) W12[i] = std::polar(1.0, 2 * M_PI * i / K2);
        for (int cell : {1, 4}) {
            std::vector<double> val(NP);
            double M = -1;
            long argM = -1;
            std::vector<cd> z((size_t)2 * K1 * K2), u((size_t)2 * K1 * K2);
            for (int t3 = 0; t3 < K3; ++t3) {
                for (int comp = 0; comp < 2; ++comp)
                    for (int k1 = 0; k1 < K1; ++k1)
                        for (int k2 = 0; k2 < K2; ++k2)
                            z[((size_t)comp * K1 + k1) * K2 + k2] = unpack(Bw[(size_t)cell * kCellElems + bw_piece(k1 / 2, k2, t3) + comp * 2 + (k1 & 1)]);
                for (int comp = 0; comp < 2; ++comp)  // 53 points over k1
                    for (int t1 = 0; t1 < K1; ++t1)
                        for (int k2 = 0; k2 < K2; ++k2) {
                            cd a = 0;
                            for (int k1 = 0; k1 < K1; ++k1) a += z[((size_t)comp * K1 + k1) * K2 + k2] * W53[(k1 * t1) % K1];
                            u[((size_t)comp * K1 + t1) * K2 + k2] = a;
                        }
                for (int t1 = 0; t1 < K1; ++t1)
                    for (int t2 = 0; t2 < K2; ++t2) {
                        cd yd = 0, yp = 0;
                        for (int k2 = 0; k2 < K2; ++k2) yd += u[((size_t)0 * K1 + t1) * K2 + k2] * W12[(k2 * t2) % K2], yp += u[((size_t)1 * K1 + t1) * K2 + k2] * W12[(k2 * t2) % K2];
                        const double v = w0 * std::abs(yd) + w1 * std::abs(yp);
                        const long t = lag_of(t1, t2, t3);
                        val[t] = v;
                        if (v > M || (v == M && t < argM)) M = v, argM = t;
                    }
            }
            float gv;
            { const unsigned gb = (unsigned)(cm[cell] >> 32); memcpy(&gv, &gb, 4); }
            const long glag = (long)(~(unsigned)(cm[cell] & 0xffffffffu));
            long want = 0, found = 0;
            for (long t = 0; t < NP; ++t)
                if (val[t] >= M * (keep + 2e-6)) {  // (the kernel's fp32 values are ~1e-6 from these)
                    ++want;
                    for (const auto &e : ex)
                        if (e.cell == cell && e.lag == t) {
                            ++found;
                            break;
                        }
                }
            printf("sieve protocol, cell %d: maximum %.6g at lag %ld (float64 of the same buffer: %.6g at %ld; value error %.1e); %ld lags within the tolerance of it, %ld of them on the list (%d entries in all)\n",
                   cell, gv, glag, M, argM, std::fabs(gv - M) / M, want, found, n_ex);
        }
    }
    printf("column pass: worst error of |y|^2 against a float64 53 x 12 transform of the SAME buffer: %.3g of the largest (%.3g)\n", worst_cols / big, big);
    printf("pair, end to end: worst relative error of |y| against the N-point sum in natural order (index maps, rotation, fp16 storage): %.2e\n", worst_e2e);

    // ---------------- timing at cfg3 sizes: prns x 201 cells, row workgroups walk all 201 bins of their PRN
    for (int p = 0; p < prns; ++p)
        for (int b = 0; b < D; ++b) h_bin[p * D + b] = b, h_cs[p * D + b] = (long)(p % nslots) * 2 * NP;
    CK(hipMemcpy(d_bin, h_bin.data(), ncells * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cs, h_cs.data(), ncells * sizeof(long), hipMemcpyHostToDevice));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));
    for (int gc : {67}) {
        RowsArgs rt{d_Xs, d_Cs, d_Bw, d_bin, d_cs, ncells, gc, 1};
        const int chunks = (ncells + gc - 1) / gc;
        for (int cgrid : {4096, 8192})
        for (int qch : {4, 1}) {
            float best_r = 1e9f, best_c = 1e9f;
            for (int rep = 0; rep <= reps; ++rep) {  // the last repetition counts the passes (one atomic per wave item: not timed)
                const bool counting = rep == reps;
                CK(hipMemset(d_cellmax, 0, ncells * sizeof(unsigned long long)));
                CK(hipMemset(d_lb, 0, ncells * sizeof(float)));
                CK(hipMemset(d_extra_count, 0, sizeof(int)));
                CK(hipMemset(d_stats, 0, 4 * sizeof(unsigned long long)));
                ColsArgs ct{d_Bw, d_coef, ncells, 0.52440442f, 0.85146932f, d_cellmax, d_lb, D, d_extra, d_extra_count, extra_cap, 0, 0.996f, qch, counting ? d_stats : nullptr, nullptr, -1, -1};
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_pfa_rows<2>, dim3(MP * K2 * chunks), dim3(kRowsThreads), rows_lds, 0, rt);
                CK(hipEventRecord(e1));
                hipLaunchKernelGGL((k_pfa_cols<2, false>), dim3(cgrid), dim3(kColsThreads), kColsLds, 0, ct);
                CK(hipEventRecord(e2));
                CK(hipDeviceSynchronize());
                float mr, mc;
                CK(hipEventElapsedTime(&mr, e0, e1));
                CK(hipEventElapsedTime(&mc, e1, e2));
                if (!counting) best_r = std::min(best_r, mr), best_c = std::min(best_c, mc);
            }
            unsigned long long st[4];
            int n_ex = 0;
            CK(hipMemcpy(st, d_stats, sizeof(st), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&n_ex, d_extra_count, sizeof(int), hipMemcpyDeviceToHost));
            printf("timing: %d PRNs x %d bins, %d-cell row workgroups, column grid %d, %d-block chunks: rows %.3f ms + columns %.3f ms per 201 cells = %.3f ms  "
                   "%llu wave items, %llu through the values' pass (%.2f %%; %.2f of 7 output blocks each), %llu exhaustive, %d list entries\n",
                   prns, D, gc, cgrid, qch, best_r / prns, best_c / prns, (best_r + best_c) / prns, st[0], st[1], 100.0 * st[1] / std::max(1ull, st[0]),
                   (double)st[3] / std::max(1ull, st[1]), st[2], n_ex);
        }
    }
    // ---------------- the row pass of one half of the cells beside the column pass of the other half (two streams), against the same four launches in a row
    if (prns >= 2) {
        hipStream_t s1, s2;
        CK(hipStreamCreate(&s1));
        CK(hipStreamCreate(&s2));
        hipEvent_t evA, evB, t0, t1;
        CK(hipEventCreate(&evA));
        CK(hipEventCreate(&evB));
        CK(hipEventCreate(&t0));
        CK(hipEventCreate(&t1));
        const int nA = (prns / 2) * D, nB = ncells - nA, gc = 67;
        auto rows = [&](hipStream_t st, int off, int n) {
            RowsArgs r{d_Xs, d_Cs, d_Bw + (size_t)off * kCellElems, d_bin + off, d_cs + off, n, gc, 1};
            hipLaunchKernelGGL(k_pfa_rows<2>, dim3(MP * K2 * ((n + gc - 1) / gc)), dim3(kRowsThreads), rows_lds, st, r);
        };
        auto cols = [&](hipStream_t st, int off, int n) {
            ColsArgs c{d_Bw + (size_t)off * kCellElems, d_coef, n, 0.52440442f, 0.85146932f, d_cellmax, d_lb, D, d_extra, d_extra_count, extra_cap, off, 0.996f, 1, nullptr, nullptr, -1, -1};
            hipLaunchKernelGGL((k_pfa_cols<2, false>), dim3(8192), dim3(kColsThreads), kColsLds, st, c);
        };
        float best_seq = 1e9f, best_ovl = 1e9f;
        for (int rep = 0; rep < 2 * reps; ++rep) {
            const bool ovl = rep & 1;
            CK(hipMemset(d_cellmax, 0, ncells * sizeof(unsigned long long)));
            CK(hipMemset(d_lb, 0, ncells * sizeof(float)));
            CK(hipMemset(d_extra_count, 0, sizeof(int)));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, s1));
            rows(s1, 0, nA);
            CK(hipEventRecord(evA, s1));
            hipStream_t sc = ovl ? s2 : s1;
            if (ovl) { CK(hipStreamWaitEvent(s2, evA, 0)); }
            if (ovl) {
                rows(s1, nA, nB);  // (enqueued first: both queues have work when the first row pass ends)
                CK(hipEventRecord(evB, s1));
                cols(sc, 0, nA);
                CK(hipStreamWaitEvent(s2, evB, 0));
                cols(sc, nA, nB);
                CK(hipEventRecord(t1, s2));
            } else {
                cols(s1, 0, nA);
                rows(s1, nA, nB);
                cols(s1, nA, nB);
                CK(hipEventRecord(t1, s1));
            }
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, t0, t1));
            (ovl ? best_ovl : best_seq) = std::min(ovl ? best_ovl : best_seq, ms);
        }
        printf("two halves of %d + %d cells: rows, columns, rows, columns in a row %.3f ms per 201 cells; the second row pass beside the first column pass (two streams) %.3f ms\n",
               nA, nB, best_seq / prns, best_ovl / prns);
    }
    return 0;
}
