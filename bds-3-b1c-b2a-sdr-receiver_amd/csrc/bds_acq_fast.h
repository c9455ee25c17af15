// Specialised search kernels for the hot transform plans (compile-time lengths; see bds_fft_t.h).
// Same math, same HBM layout and same outputs as k_rows_inv / k_cols_inv_max in
// bds_acq_kernels.h -- the generic kernels remain the fallback for every other length.
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

template <int S>
__host__ __device__ constexpr int rows_threads() { return S / 16 < 64 ? 64 : ((S / 16 + 63) / 64) * 64; }
template <int S>
__host__ __device__ constexpr int cols_threads() { return S / 2; }  // T = 8 columns, 16 points per thread
constexpr int kFastT = 8;

// ---- forward row pass with a typed store (k_rows_fwd of bds_acq_kernels.h, run-time plan) --------
template <class ST>
__global__ __launch_bounds__(1024) void k_rows_fwd_st(Plan1D p, const float2 *__restrict__ in, long in_stride,
                                                      ST *__restrict__ out, long out_stride, int conj_flag,
                                                      float scale) {
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
        st_c(dst, e, v);
    }
}

// ---- inverse row pass ------------------------------------------------------------------------
// 1-D grid of L1*G workgroups.  Workgroups that handle the same spectrum row k1 for the G
// Doppler bins of a launch are consecutive on ONE XCD (hardware places workgroup b on XCD b % 8),
// so the code-spectrum rows are fetched from HBM once and re-used out of that XCD's L2.
template <int S, int NCOMP, class ST>
__global__ __launch_bounds__(rows_threads<S>(), 3) void k_rows_inv_t(const float2 *__restrict__ tw, TwiddleL twl,
                                                                const ST *__restrict__ Xs, long L, int L1, int G,
                                                                int bin0, const ST *__restrict__ Cs,
                                                                ST *__restrict__ Bw, float out_scale) {
    constexpr int NT = rows_threads<S>();
    constexpr int PE = (S + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];  // tspan<S>() elements
    __shared__ float2 s_step[PE];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int g = m % G, k1 = (m / G) * 8 + xcd;
    if (k1 >= L1) return;
    if (tid < PE) {
        const long mm = (long)k1 * ((long)tid * NT);
        s_step[tid] = mm < L ? twl.get<+1>((uint32_t)mm) : make_float2(1.f, 0.f);
    }
    const float2 wbase = tid < S ? twl.get<+1>((uint32_t)k1 * (uint32_t)tid) : make_float2(1.f, 0.f);
    const ST *xr = Xs + (long)(bin0 + g) * L + (long)k1 * S;
    float2 xv[PE];
#pragma unroll
    for (int i = 0; i < PE; ++i) {
        const int e = tid + i * NT;
        if (S % NT == 0 || e < S) xv[i] = ld_c(xr, e);
    }
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
        const ST *cr = Cs + (long)comp * L + (long)k1 * S;
#pragma unroll
        for (int i = 0; i < PE; ++i) {
            const int e = tid + i * NT;
            if (S % NT == 0 || e < S) lds[e + (e >> 4)] = cmul(xv[i], ld_c(cr, e));
        }
        __syncthreads();
        TPlan<S>::template run<1, NT, +1>(lds, tw, tid);
        ST *dst = Bw + ((long)g * NCOMP + comp) * L + (long)k1 * S;
#pragma unroll
        for (int i = 0; i < PE; ++i) {
            const int e = tid + i * NT;
            if (S % NT == 0 || e < S) {
                float2 y = cmul(lds[e + (e >> 4)], cmul(wbase, s_step[i]));
                y.x *= out_scale;
                y.y *= out_scale;
                st_c(dst, e, y);
            }
        }
        __syncthreads();
    }
}

// ---- inverse column pass + |.| combine + maximum -----------------------------------------------
// grid (tiles, cells), T = 8 columns per workgroup.  The second component's tile is fetched into
// registers while the first one is being transformed.
template <int S, int NCOMP, class ST>
__global__ __launch_bounds__(cols_threads<S>(), (2 * cols_threads<S>() + 255) / 256) void k_cols_inv_max_t(const float2 *__restrict__ tw, int L2,
                                                                     const ST *__restrict__ Bw, long L, float w0,
                                                                     float w1, int lo1, int hi1, int lo2, int hi2,
                                                                     Rec *__restrict__ recs, int rec_stride) {
    constexpr int NT = cols_threads<S>();
    constexpr int T = kFastT;
    constexpr int SP = tspan<S>();
    constexpr int NI = 8;  // (row, column-pair) items per thread: S*4 / NT
    extern __shared__ __attribute__((aligned(16))) float2 lds[];  // T * SP elements
    __shared__ float s_v[NT / 64];
    __shared__ int s_l[NT / 64];
    const int tid = threadIdx.x;
    const int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int g = blockIdx.y;
    const int c0 = tile * T;
    float2 pre[NI][2];
    auto fetch = [&](int comp) {
        const ST *src = Bw + ((long)g * NCOMP + comp) * L;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + i * NT;
            const int r = it >> 2, cp = (it & 3) * 2;
            const long o = (long)r * L2 + c0 + cp;
            if (c0 + cp + 1 < L2) {
                pre[i][0] = ld_c(src, o);
                pre[i][1] = ld_c(src, o + 1);
            } else {
                pre[i][0] = c0 + cp < L2 ? ld_c(src, o) : make_float2(0.f, 0.f);
                pre[i][1] = make_float2(0.f, 0.f);
            }
        }
    };
    fetch(0);
    float mag[NI][2];
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + i * NT;
            const int r = it >> 2, cp = (it & 3) * 2;
            const int pr = r + (r >> 4);
            lds[cp * SP + pr] = pre[i][0];
            lds[(cp + 1) * SP + pr] = pre[i][1];
        }
        __syncthreads();
        if (comp + 1 < NCOMP) fetch(comp + 1);  // in flight during the transform
        TPlan<S>::template run<T, NT, +1>(lds, tw, tid);
        const float w = comp == 0 ? w0 : w1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + i * NT;
            const int r = it >> 2, cp = (it & 3) * 2;
            const int pr = r + (r >> 4);
            const float2 a = lds[cp * SP + pr], b = lds[(cp + 1) * SP + pr];
            const float ma = w * sqrtf(a.x * a.x + a.y * a.y), mb = w * sqrtf(b.x * b.x + b.y * b.y);
            mag[i][0] = comp == 0 ? ma : mag[i][0] + ma;
            mag[i][1] = comp == 0 ? mb : mag[i][1] + mb;
        }
        __syncthreads();
    }
    float bv = -1.f;
    int bl = -1;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int it = tid + i * NT;
        const int r = it >> 2, cp = (it & 3) * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = c0 + cp + h;
            const long lag = (long)r * L2 + col;
            const bool in = col < L2 && ((lag >= lo1 && lag <= hi1) || (lag >= lo2 && lag <= hi2));
            if (in) rec_better(bv, bl, mag[i][h], (int)lag);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(bv, off, 64);
        const int ol = __shfl_down(bl, off, 64);
        rec_better(bv, bl, ov, ol);
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) {
        s_v[wave] = bv;
        s_l[wave] = bl;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w2 = 1; w2 < NT / 64; ++w2) rec_better(bv, bl, s_v[w2], s_l[w2]);
        Rec rr;
        rr.v = bv;
        rr.lag = bl;
        recs[(long)g * rec_stride + tile] = rr;
    }
}

}  // namespace bds
