// Forward transforms and shared helpers for the hot transform plans (compile-time lengths; see bds_fft_t.h).
// The search kernels of these plans live in bds_acq_f32.h (row pass, tile column pass) and bds_acq_wcols.h (wave-private
// column pass); the run-time-plan kernels of bds_acq_kernels.h remain the fallback for every other length.
// (The packed-fp16 search arithmetic of rounds 1-2 -- k_rows_inv_h / k_cols_inv_max_h and their fused launch chain -- is
// retired: with the wave-private column pass the fp32-arithmetic pair runs within 20 % of it, it never earned a headline, and
// its one-record-per-tile sieve was not complete.)
//
// ST = storage type of the spectra and of the inter-pass buffer:
//   float2  : fp32 complex (8 B)
//   __half2 : fp16 complex (4 B) -- halves the HBM traffic of the search; the values are
//             pre-scaled by powers of two on the host so they sit mid-range, all arithmetic
//             stays fp32, and the f64 refinement makes the final decision either way.
#pragma once

#include <hip/hip_fp16.h>

#include "bds_acq_kernels.h"
#include "bds_fft_t.h"

namespace bds {

__device__ __forceinline__ float2 ld_c(const float2 *p, long i) { return p[i]; }
__device__ __forceinline__ float2 ld_c(const __half2 *p, long i) { return __half22float2(p[i]); }
__device__ __forceinline__ void st_c(float2 *p, long i, float2 v) { p[i] = v; }
__device__ __forceinline__ void st_c(__half2 *p, long i, float2 v) { p[i] = __float22half2_rn(v); }

// Element order of a 4096-point spectrum row as the wave-private row pass reads it (bds_acq_wrows.h, round 4): thread t of
// its workgroup -- wave w = t / 64, lane (ql = t & 3, bl = (t & 63) >> 2), q' = 4 w + ql -- multiplies the sixteen elements
// e = 16 bl + q' + 256 bh, bh = 0 .. 15.  Stored at [16 t + bh] they are 64 contiguous bytes per thread and 4 KB per wave:
// four 16-byte loads per lane and row instead of sixteen 4-byte ones in 16-byte pieces.  The forward row pass writes the rows
// of the signal and code spectra in this order whenever the search will run that kernel (spectra_permuted() in bds_acq.hip).
__host__ __device__ __forceinline__ constexpr int wrows_perm(int e) {
    const int bh = e >> 8, r = e & 255, bl = r >> 4, qp = r & 15;
    const int t = 64 * (qp >> 2) + 4 * bl + (qp & 3);
    return 16 * t + bh;
}

template <int S>
__host__ __device__ constexpr int rows_threads() { return S / 16 < 64 ? 64 : ((S / 16 + 63) / 64) * 64; }
#ifndef BDS_COLS768X8_NT
#define BDS_COLS768X8_NT 512
#endif
template <int S, int T>
__host__ __device__ constexpr int cols_threads() {
    // 16 points per thread; 768 x 4 takes 256 so that its radix-3 stage (256 butterflies per column)
    // maps one column per unroll step with no index arithmetic (the radix-16 stages idle one wave)
    // (768 x 8 likewise takes 512: three items per thread, the occupancy of the 4-column kernel, and two
    // adjacent lanes share each 32-byte piece of a tile row)
    if (S == 768) return T == 4 ? 256 : BDS_COLS768X8_NT;
    return S * T / 16;
}

// ---- forward row pass with a typed store (k_rows_fwd of bds_acq_kernels.h, run-time plan) --------
template <class ST>
__global__ __launch_bounds__(1024) void k_rows_fwd_st(Plan1D p, const float2 *__restrict__ in, long in_stride,
                                                      ST *__restrict__ out, long out_stride, int conj_flag,
                                                      float scale, int perm) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int row = blockIdx.x, batch = blockIdx.y;
    const int S = p.S;
    const float2 *src = in + (long)batch * in_stride + (long)row * S;
    for (int e = tid; e < S; e += nthr) lds[lds_phys(e)] = src[e];
    __syncthreads();
    fft_lds<-1>(lds, p, S, 1, tid, nthr);
    ST *dst = out + (long)batch * out_stride + (long)row * S;
    for (int e = tid; e < S; e += nthr) {
        float2 v = lds[lds_phys(e)];
        v.x *= scale;
        v.y *= conj_flag ? -scale : scale;
        st_c(dst, perm ? wrows_perm(e) : e, v);  // (perm: 4096-point rows only)
    }
}

// exp(-j 2 pi x), 0 <= x < 1 given in f64: fp32 sine / cosine of the fp32 part of x, corrected to first order for its rounding
__device__ __forceinline__ void phasor32_neg(double x, float *c, float *s) {
    const float hi = (float)x;
    const float lo = (float)(x - (double)hi);
    float s0, c0;
    sincospif(2.0f * hi, &s0, &c0);
    const float d = 6.28318530717958647692f * lo;
    *c = c0 - d * s0;
    *s = -(s0 + d * c0);
}

template <class L, class = void>
struct has_carrier : std::false_type {};
template <class L>
struct has_carrier<L, std::enable_if_t<L::kHasCarrier>> : std::true_type {};

// ---- forward passes on the compile-time stages (fp32 arithmetic) -------------------------------
// Same two passes as k_cols_fwd / k_rows_fwd_st above with the transform lengths as template
// parameters.  Column pass: T columns per workgroup (4: measured 4.3 ms per 201 bins against 4.7 ms with
// 8 and 5.8 ms on the run-time engine); the loader (carrier wipe-off of the int8 block by a phasor
// rotation, or the sampled code) feeds the tile straight into LDS.
template <int S, int T, class Loader>
__global__ __launch_bounds__((cols_threads<S, T>()), 2) void k_cols_fwd_t(const float2 *__restrict__ tw, TwiddleL twl,
                                                                        int L2, Loader ld,
                                                                        float2 *__restrict__ out, long out_stride) {
    constexpr int NT = cols_threads<S, T>();
    constexpr int SP = tspan<S>();
    extern __shared__ __attribute__((aligned(16))) float2 lds[];  // T * SP data + twiddle table
    float2 *tw_lds = lds + T * SP;
    const int tid = threadIdx.x;
    load_twiddles<S, NT>(tw_lds, tw, tid);
    const int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int batch = blockIdx.y;
    const int c0 = tile * T;
    if constexpr (has_carrier<Loader>::value) {
        // a thread keeps its column and walks the rows in steps of NT / T: the carrier advances by a constant angle, so it is
        // a phasor rotated in fp32 from a start taken at the f64-reduced phase (SignalLoader::phasor32), taken again where the
        // periodic extension wraps (the phase index restarts there) and every 16th step (S * T / NT = 12 steps here: never).
        // Rounds 1-4 rotated in f64 and called the f64 sincospi five times per thread -- start, step, 8th step, and twice
        // for the inter-pass twiddle below: 750 of the kernel's 1650 vector instructions per wave, and the vector pipe was
        // 100 % busy (profiles/r04_b1c_pmc.txt).  fp32 rotation over <= 15 steps: ~1e-6, the rounding of the transform itself.
        static_assert(NT % T == 0, "column of a thread must stay fixed");
        const int j = tid % T, col = c0 + j;
        const long dn = (long)(NT / T) * L2;
        float wr, wi, c = 1.f, sn = 0.f;
        ld.step32(batch, dn, &wr, &wi);
        int it = 0;
        for (int r = tid / T; r < S; r += NT / T, ++it) {
            const long n = (long)r * L2 + col;
            if ((it & 15) == 0 || (n >= ld.n_circ && n - dn < ld.n_circ)) {
                ld.exact32(batch, n, &c, &sn);
            } else {
                const float nr = c * wr - sn * wi;
                sn = c * wi + sn * wr;
                c = nr;
            }
            lds[j * SP + r + (r >> 4)] = col < L2 ? ld.mix(n, c, sn) : make_float2(0.f, 0.f);
        }
    } else {
        for (int e = tid; e < S * T; e += NT) {
            const int r = e / T, j = e % T;
            const int col = c0 + j;
            lds[j * SP + r + (r >> 4)] = col < L2 ? ld(batch, (long)r * L2 + col) : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    TPlan<S>::template run<T, NT, -1>(lds, tw_lds, tid, LdsIO{}, LdsIO{});
    float2 *o = out + (long)batch * out_stride;
    // inter-pass twiddle W_L^(k1 col) = exp(-j 2 pi k1 col / L): a thread keeps its column and walks k1
    // in steps of NT / T, so the twiddle is a rotation from a start whose phase is an exact integer ratio (a table look-up
    // per element is two scattered loads per lane and made this kernel TA-bound); fp32 phasors as for the carrier above
    static_assert(NT % T == 0, "column of a thread must stay fixed");
    (void)twl;
    {
        const int j = tid % T, col = c0 + j, k0 = tid / T;
        constexpr int dk = NT / T;
        const long L = (long)S * L2;
        float wr, wi, c, sn;
        phasor32_neg((double)(((long)dk * col) % L) / (double)L, &wr, &wi);
        phasor32_neg((double)(((long)k0 * col) % L) / (double)L, &c, &sn);
        int it = 0;
        for (int k1 = k0; k1 < S; k1 += dk, ++it) {
            if (it && (it & 15) == 0) phasor32_neg((double)(((long)k1 * col) % L) / (double)L, &c, &sn);
            if (col < L2) o[(long)k1 * L2 + col] = cmul(lds[j * SP + k1 + (k1 >> 4)], make_float2(c, sn));
            const float nr = c * wr - sn * wi;
            sn = c * wi + sn * wr;
            c = nr;
        }
    }
}

// Row pass: first stage straight from the inter-pass buffer, last stage straight to the typed store
// (conj / scale as k_rows_fwd_st).
template <int S, class ST>
__global__ __launch_bounds__(rows_threads<S>(), 3) void k_rows_fwd_t(const float2 *__restrict__ tw,
                                                                   const float2 *__restrict__ in, long in_stride,
                                                                   ST *__restrict__ out, long out_stride,
                                                                   int conj_flag, float scale, int perm) {
    constexpr int NT = rows_threads<S>();
    extern __shared__ __attribute__((aligned(16))) float2 lds[];  // tspan<S>() data + twiddle table
    float2 *tw_lds = lds + tspan<S>();
    const int tid = threadIdx.x;
    load_twiddles<S, NT>(tw_lds, tw, tid);
    const int row = blockIdx.x, batch = blockIdx.y;
    const float2 *src_row = in + (long)batch * in_stride + (long)row * S;
    ST *dst = out + (long)batch * out_stride + (long)row * S;
    const float sy = conj_flag ? -scale : scale;
    auto src = [&](int, int, int, int e) { return src_row[e]; };
    // perm (4096-point rows, fp16 storage): the last stage leaves thread t the outputs t + 256 q, which wrows_perm() puts at
    // sixteen consecutive stored elements -- collected in registers, then written below
    constexpr bool kCanPerm = S == 4096 && std::is_same<ST, __half2>::value;
    [[maybe_unused]] uint32_t hv[16];
    auto dstf = [&](int, int q, int, int e, float2 v) {
        const float2 w = make_float2(v.x * scale, v.y * sy);
        if constexpr (kCanPerm) {
            if (perm) {
                const __half2 h = __float22half2_rn(w);
                hv[q] = *reinterpret_cast<const uint32_t *>(&h);
                return;
            }
        }
        st_c(dst, e, w);
    };
    TPlan<S>::template run<1, NT, -1>(lds, tw_lds, tid, src, dstf);
    if constexpr (kCanPerm) {
        if (perm) {
            // through LDS, so that the row leaves in full lines (a thread's 64 bytes straight to memory are four 16-byte
            // pieces per store instruction at a 64-byte stride: forward pass 3.7 -> 4.05 ms)
            __syncthreads();  // every thread has read its inputs of the last stage
            uint4 *l4 = reinterpret_cast<uint4 *>(lds);
#pragma unroll
            for (int k = 0; k < 4; ++k) l4[wrows_perm(tid) / 4 + k] = make_uint4(hv[4 * k], hv[4 * k + 1], hv[4 * k + 2], hv[4 * k + 3]);
            __syncthreads();
            uint4 *g4 = reinterpret_cast<uint4 *>(dst);
#pragma unroll
            for (int k = 0; k < 4; ++k) g4[tid + NT * k] = l4[tid + NT * k];
        }
    }
}

// 16-byte-per-lane global access: four consecutive complex elements
struct C4 {
    float2 v[4];
};
__device__ __forceinline__ C4 ld4(const float2 *p, long i) {
    const float4 *q = reinterpret_cast<const float4 *>(p + i);
    const float4 a = q[0], b = q[1];
    C4 r;
    r.v[0] = make_float2(a.x, a.y);
    r.v[1] = make_float2(a.z, a.w);
    r.v[2] = make_float2(b.x, b.y);
    r.v[3] = make_float2(b.z, b.w);
    return r;
}
__device__ __forceinline__ C4 ld4(const __half2 *p, long i) {
    union {
        uint4 u;
        __half2 h[4];
    } t;
    t.u = *reinterpret_cast<const uint4 *>(p + i);
    C4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = __half22float2(t.h[k]);
    return r;
}
__device__ __forceinline__ void st4(float2 *p, long i, const C4 &c) {
    float4 *q = reinterpret_cast<float4 *>(p + i);
    q[0] = make_float4(c.v[0].x, c.v[0].y, c.v[1].x, c.v[1].y);
    q[1] = make_float4(c.v[2].x, c.v[2].y, c.v[3].x, c.v[3].y);
}
__device__ __forceinline__ void st4(__half2 *p, long i, const C4 &c) {
    union {
        uint4 u;
        __half2 h[4];
    } t;
#pragma unroll
    for (int k = 0; k < 4; ++k) t.h[k] = __float22half2_rn(c.v[k]);
    *reinterpret_cast<uint4 *>(p + i) = t.u;
}

template <int S>
struct PlanInfo {
    static constexpr int kN = sizeof(TPlan<S>::kRadix) / sizeof(int);
    static constexpr int kLast = TPlan<S>::kRadix[kN - 1];  // radix of the last stage
    static constexpr int kNsLast = S / kLast;               // its NS (= its butterfly count per transform)
};

}  // namespace bds