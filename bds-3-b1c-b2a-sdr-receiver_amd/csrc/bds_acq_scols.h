// Inverse column pass for SHORT columns (gfx950): the 80 x 4096 plan of small searches (B2a at 99.375 MS/s: N + X - 1 =
// 298 124 <= 327 680 = 80 x 4096; round 4).
//
// Why this plan: the 256 x 1280 plan such a search used (rounds 1-3) leaves the row pass on the round-2 kernel (1280 is not a
// power of 16).  80 x 4096 has the same transform length, its rows are the 4096-point rows the wave-private row pass
// (bds_acq_wrows.h) is built for, and its columns are so short that ONE LANE transforms a whole column in registers:
//
//   k1 = 5 m + r (m < 16, r < 5),  n1 = d + 16 c (d < 16, c < 5):
//   y[d + 16 c] = sum_r w5^(r c) w80^(r d) Y_r[d],      Y_r[d] = sum_m w16^(m d) B[5 m + r]          (inverse direction)
//
// five 16-point transforms (packed fp32, bds_fft_pk.h), 64 twiddle products, sixteen 5-point transforms of which only the
// outputs c = 0, 1, 2 are formed (the searched lags n1 L2 + n2 < N end inside output row 48: n1 <= 48 = 0 + 16 * 3, so c = 3 is
// needed for d = 0 alone).  No LDS in the transform, no barrier, no lane-dependent constant: every twiddle is the same for all
// lanes and comes from scalar registers / scalar loads.  Lanes 2 i and 2 i + 1 take the two components of column c0 + i (the
// inter-pass buffer holds them side by side, k_rows_wave_f<2, true>): a wave-instruction of the 80 row loads reads 256 contiguous
// bytes, the partner's |y|^2 arrives by one DPP move.  ~180 VGPRs: two waves per SIMD, which a stream of packed instructions
// without waits fills (tools/probe/coissue.hip).
//
// Outputs exactly as the wave-private column pass (bds_acq_wcols.h): per cell the packed maximum {value, first lag} by atomic
// max, per PRN the running bound lb, the candidate list -- same completeness argument, same host code.
#pragma once

#include "bds_acq_wcols.h"

namespace bds {

#ifndef BDS_SCOLS_OCC
#define BDS_SCOLS_OCC 2
#endif
constexpr int kSColsLen = 80;      // column length
constexpr int kSColsOut = 49;      // output rows formed: n1 = 0 .. 48
constexpr int kSColsNT = 256;      // threads: 128 columns x 2 components
constexpr size_t kSColsLdsBytes = sizeof(float) * kSColsOut * kSColsNT;  // staging of the (rare) exact-value tail

struct SColsArgs {
    const float2 *tw80;  // w80^k, k = 0 .. 79, inverse direction (exp(+2 pi j k / 80))
    int L2, G;
    const void *Bw;      // [cell][k1][n2][component] fp16 complex
    long L;
    float w0, w1;
    int lo1, hi1, lo2, hi2;
    const int4 *cell_rng;            // optional per-cell (lo1, hi1, lo2, hi2)
    unsigned long long *cellmax;     // as WColsArgs
    float *lb;
    int lb_div;
    Extra *extra;
    int *extra_count;
    int extra_cap;
    int cell0;
    float keep;
    const int *cell_src;             // optional: cell g's rows lie at cell index cell_src[g] of Bw (the B2a second-peak pass reading
                                     // the winning cells straight out of the main search's inter-pass buffer), else at g
};

// 5-point inverse transform, outputs 0, 1, 2 (and 3 when WITH3): X_c = sum_r t_r w5^(r c)
template <bool WITH3>
__device__ __forceinline__ void pk_radix5_012(const v2f (&t)[5], v2f &x0, v2f &x1, v2f &x2, v2f &x3) {
    const v2f kc1 = {0.30901699437494742410f, 0.30901699437494742410f};    // cos(2 pi / 5)
    const v2f kc2 = {-0.80901699437494742410f, -0.80901699437494742410f};  // cos(4 pi / 5)
    const v2f ks1 = {0.95105651629515357212f, 0.95105651629515357212f};    // sin(2 pi / 5)
    const v2f ks2 = {0.58778525229247312917f, 0.58778525229247312917f};    // sin(4 pi / 5)
    const v2f s1 = t[1] + t[4], d1 = t[1] - t[4], s2 = t[2] + t[3], d2 = t[2] - t[3];
    x0 = t[0] + s1 + s2;
    const v2f r1 = __builtin_elementwise_fma(s2, kc2, __builtin_elementwise_fma(s1, kc1, t[0]));
    const v2f r2 = __builtin_elementwise_fma(s2, kc1, __builtin_elementwise_fma(s1, kc2, t[0]));
    const v2f q1 = __builtin_elementwise_fma(d2, ks2, d1 * ks1);   // s1 d1 + s2 d2
    const v2f q2 = __builtin_elementwise_fma(d2, -ks1, d1 * ks2);  // s2 d1 - s1 d2
    x1 = pk_addj(r1, q1);  // inverse: X1 = r1 + j q1
    x2 = pk_addj(r2, q2);
    if constexpr (WITH3) x3 = pk_subj(r2, q2);
}

template <int NCOMP>
__global__ __launch_bounds__(kSColsNT, BDS_SCOLS_OCC) void k_cols_small_f(SColsArgs A) {
    static_assert(NCOMP == 2, "two components side by side in the inter-pass buffer");
    extern __shared__ __attribute__((aligned(16))) float sm_all[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L2 = A.L2;
    const int tiles = L2 / (kSColsNT / 2);
    // consecutive workgroups work on different cells (a cell's running maximum is then settled by a few early waves, as in the
    // wave-private column pass); a workgroup reads 80 x 1 KB of whole lines, so no two share a line
    const int item = (int)blockIdx.x;
    const int g = item % A.G, tile = item / A.G;
    (void)tiles;
    const int comp = tid & 1, col = tile * (kSColsNT / 2) + (tid >> 1);
    const long gs = A.cell_src ? A.cell_src[g] : g;
    const uint32_t *src = (const uint32_t *)A.Bw + (gs * A.L + col) * 2 + comp;
    uint32_t in[kSColsLen];
#pragma unroll
    for (int k1 = 0; k1 < kSColsLen; ++k1) in[k1] = src[(long)k1 * L2 * 2];
    const int cell = A.cell0 + g;
    float *const lbp = A.lb + cell / A.lb_div;
    const float lbv = __hip_atomic_load(lbp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned cur = (unsigned)(__hip_atomic_load(A.cellmax + cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);

    // ---- Y_r[d] = sum_m w16^(m d) B[5 m + r]
    v2f Y[5][16];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 t = h2_to_f2(in[5 * m + r]);
            Y[r][m] = (v2f){t.x, t.y};
        }
        pk_bfly16<false>(Y[r], nullptr);
    }
    // ---- y[d + 16 c] = sum_r w5^(r c) (w80^(r d) Y_r[d]); |y|^2 of this lane's component, outputs 0 .. 48
    float sq[kSColsOut];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        v2f t[5];
        t[0] = Y[0][d];
#pragma unroll
        for (int r = 1; r < 5; ++r) {
            if (d == 0) {
                t[r] = Y[r][0];
            } else {
                const float2 w = A.tw80[(r * d) % kSColsLen];  // uniform: a scalar load
                t[r] = pk_cmul_k(Y[r][d], (v2f){w.x, w.y});
            }
        }
        v2f x0, x1, x2, x3;
        if (d == 0)
            pk_radix5_012<true>(t, x0, x1, x2, x3);
        else
            pk_radix5_012<false>(t, x0, x1, x2, x3);
        sq[d] = x0.x * x0.x + x0.y * x0.y;
        sq[d + 16] = x1.x * x1.x + x1.y * x1.y;
        sq[d + 32] = x2.x * x2.x + x2.y * x2.y;
        if (d == 0) sq[48] = x3.x * x3.x + x3.y * x3.y;
    }
    // the other component's |y|^2 of the same column: the neighbouring lane (quad_perm [1, 0, 3, 2])
    float sqp[kSColsOut];
    float bmax = 0.f;
#pragma unroll
    for (int k = 0; k < kSColsOut; ++k) {
        sqp[k] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq[k]), 0xB1, 0xF, 0xF, true));
        bmax = fmaxf(bmax, sq[k] + sqp[k]);
    }
    // ---- maximum of the wave's 32 columns, candidates: as the tail of k_cols_wave_f (Cauchy-Schwarz bound first)
    const float wown = comp == 0 ? A.w0 : A.w1, wpar = comp == 0 ? A.w1 : A.w0;
    const float wsum2 = A.w0 * A.w0 + A.w1 * A.w1;
    const float bw = wave_max_f32(bmax) * wsum2 * 1.00001f;
    const float curv = __uint_as_float(cur), lim = fminf(curv, lbv * A.keep);
    if (!(bw < lim * lim)) {  // (wave-uniform; also taken while the bounds are unset or not finite)
        int lo1 = A.lo1, hi1 = A.hi1, lo2 = A.lo2, hi2 = A.hi2;
        if (A.cell_rng) {
            const int4 r = A.cell_rng[g];
            lo1 = r.x, hi1 = r.y, lo2 = r.z, hi2 = r.w;
        }
        float *sm = sm_all + wave * (kSColsOut * 64) + lane;  // [k][lane]
        float mx = -1.f;
#pragma unroll
        for (int k = 0; k < kSColsOut; ++k) {
            float a = wown * __builtin_amdgcn_sqrtf(sq[k]) + wpar * __builtin_amdgcn_sqrtf(sqp[k]);
            const int lag = k * L2 + col;
            const bool ok = comp == 0 && ((lag >= lo1 && lag <= hi1) || (lag >= lo2 && lag <= hi2));  // one lane of the pair reports
            a = ok ? a : -1.f;
            sm[k * 64] = a;
            mx = fmaxf(mx, a);
        }
        const float Mw = wave_max_f32(mx);
        if (Mw >= 0.f) {  // (wave-uniform) something of these columns is searched
            const float thr = fmaxf(Mw, lbv) * A.keep;
            const bool newmax = __float_as_uint(Mw) >= cur;
            if (newmax || __builtin_amdgcn_ballot_w64(mx >= thr) != 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                int best = 0x7fffffff, total = 0;
#pragma nounroll
                for (int k = 0; k < kSColsOut; ++k) {
                    const float a = sm[k * 64];
                    total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(a >= thr));
                    if (newmax && a == Mw) best = min(best, k * L2 + col);
                }
                if (total > 0) {  // one reservation per wave on the list's counter
                    int base = 0;
                    if (lane == 0) base = atomicAdd(A.extra_count, total);
                    base = __builtin_amdgcn_readfirstlane(base);
#pragma nounroll
                    for (int k = 0; k < kSColsOut; ++k) {
                        const float a = sm[k * 64];
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(a >= thr);
                        if (a >= thr) {
                            const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                            BDS_DASSERT(idx >= 0 && (long)k * L2 + col < A.L);
                            if (idx < A.extra_cap) {
                                Extra ex;
                                ex.v = a;
                                ex.lag = k * L2 + col;
                                ex.cell = cell;
                                A.extra[idx] = ex;
                            }
                        }
                        base += __builtin_popcountll(mask);
                    }
                }
                if (newmax) {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
                    if (lane == 0) {
                        atomicMax(A.cellmax + cell, wc_pack(Mw, best));
                        if (Mw > lbv) atomicMax(reinterpret_cast<unsigned *>(lbp), __float_as_uint(Mw));
                    }
                }
            }
        }
    }
}

}  // namespace bds
