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

template <class L, class = void>
struct has_carrier : std::false_type {};
template <class L>
struct has_carrier<L, std::enable_if_t<L::kHasCarrier>> : std::true_type {};

// ---- forward passes on the compile-time stages (fp32 arithmetic) -------------------------------
// Same two passes as k_cols_fwd / k_rows_fwd_st above with the transform lengths as template
// parameters.  Column pass: T columns per workgroup (4: measured 4.3 ms per 201 bins against 4.7 ms with
// 8 and 5.8 ms on the run-time engine); the loader (carrier wipe-off of the int8 block by an f64
// phasor rotation, or the sampled code) feeds the tile straight into LDS.
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
        // a thread keeps its column and walks the rows in steps of NT / T: the carrier advances by a
        // constant angle, so it is rotated in f64 and re-evaluated exactly every 8th step and where the
        // periodic extension wraps (the phase index restarts there)
        static_assert(NT % T == 0, "column of a thread must stay fixed");
        const int j = tid % T, col = c0 + j;
        const long dn = (long)(NT / T) * L2;
        double wr, wi, c = 1.0, sn = 0.0;
        ld.step(batch, dn, &wr, &wi);
        int it = 0;
        for (int r = tid / T; r < S; r += NT / T, ++it) {
            const long n = (long)r * L2 + col;
            if ((it & 7) == 0 || (n >= ld.n_circ && n - dn < ld.n_circ)) {
                ld.exact(batch, n, &c, &sn);
            } else {
                const double nr = c * wr - sn * wi;
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
    // in steps of NT / T, so the twiddle is an f64 rotation from an exact start (a table look-up per
    // element is two scattered loads per lane and made this kernel TA-bound)
    static_assert(NT % T == 0, "column of a thread must stay fixed");
    (void)twl;
    {
        const int j = tid % T, col = c0 + j, k0 = tid / T;
        constexpr int dk = NT / T;
        const long L = (long)S * L2;
        double wr, wi, c, sn;
        sincospi(-2.0 * (double)(((long)dk * col) % L) / (double)L, &wi, &wr);
        sincospi(-2.0 * (double)(((long)k0 * col) % L) / (double)L, &sn, &c);
        for (int k1 = k0; k1 < S; k1 += dk) {
            if (col < L2) o[(long)k1 * L2 + col] = cmul(lds[j * SP + k1 + (k1 >> 4)], make_float2((float)c, (float)sn));
            const double nr = c * wr - sn * wi;
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
                                                                   int conj_flag, float scale) {
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
    auto dstf = [&](int, int, int, int e, float2 v) { st_c(dst, e, make_float2(v.x * scale, v.y * sy)); };
    TPlan<S>::template run<1, NT, -1>(lds, tw_lds, tid, src, dstf);
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

// ---- inverse row pass ------------------------------------------------------------------------
// 1-D grid of L1*G workgroups.  Workgroups that handle the same spectrum row k1 for the G
// Doppler bins of a launch are consecutive on ONE XCD (hardware places workgroup b on XCD b % 8),
// so the code-spectrum rows are fetched from HBM once and re-used out of that XCD's L2.
// The first radix-16 stage takes its inputs (spectrum product) straight from global memory and
// the last stage stores its twiddled outputs straight back: per component the LDS sees two
// stage hand-offs instead of four, and four barriers instead of eight.
template <int S, int NCOMP, class ST>
__global__ __launch_bounds__(rows_threads<S>(), 3) void k_rows_inv_t(const float2 *__restrict__ tw, TwiddleL twl,
                                                                   const ST *__restrict__ Xs, long L, int L1, int G,
                                                                   int bin0, const ST *__restrict__ Cs,
                                                                   ST *__restrict__ Bw, float out_scale) {
    constexpr int NT = rows_threads<S>();
    constexpr int NB1 = S / 16;                       // first-stage butterflies
    constexpr int MB1 = (NB1 + NT - 1) / NT;
    constexpr int RL = PlanInfo<S>::kLast, NSL = PlanInfo<S>::kNsLast;
    constexpr int MBL = (NSL + NT - 1) / NT;          // last-stage butterflies per thread
    extern __shared__ __attribute__((aligned(16))) float2 lds[];  // tspan<S>() data + twiddle table
    __shared__ float2 s_a[MBL], s_b[RL];  // W^(k1*NT*i), W^(k1*NSL*q)
    float2 *tw_lds = lds + tspan<S>();
    const int tid = threadIdx.x;
    load_twiddles<S, NT>(tw_lds, tw, tid);
    const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int g = m % G, k1 = (m / G) * 8 + xcd;
    if (k1 >= L1) return;
    if (tid < MBL) s_a[tid] = twl.get<+1>((uint32_t)((long)k1 * NT * tid));
    if (tid >= 64 && tid < 64 + RL) s_b[tid - 64] = twl.get<+1>((uint32_t)((long)k1 * NSL * (tid - 64)));
    const float2 wbase = twl.get<+1>((uint32_t)k1 * (uint32_t)tid);  // tid < NT <= S
    const ST *xr = Xs + (long)(bin0 + g) * L + (long)k1 * S;
    float2 xv[MB1][16];
#pragma unroll
    for (int i = 0; i < MB1; ++i) {
        const int bb = tid + i * NT;
        if (NB1 % NT == 0 || bb < NB1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) xv[i][q] = ld_c(xr, bb + q * NB1);
        }
    }
    __syncthreads();  // twiddle tables + s_a/s_b visible
    float2 wi[MBL];
#pragma unroll
    for (int i = 0; i < MBL; ++i) {
        wi[i] = cmul(wbase, s_a[i]);
        wi[i].x *= out_scale;
        wi[i].y *= out_scale;
    }
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
        const ST *cr = Cs + (long)comp * L + (long)k1 * S;
        ST *dst = Bw + ((long)g * NCOMP + comp) * L + (long)k1 * S;
        auto src = [&](int i, int q, int, int e) { return cmul(xv[i][q], ld_c(cr, e)); };
        auto out = [&](int i, int q, int, int e, float2 v) { st_c(dst, e, cmul(v, cmul(wi[i], s_b[q]))); };
        TPlan<S>::template run<1, NT, +1>(lds, tw_lds, tid, src, out);
        if (comp + 1 < NCOMP) __syncthreads();  // last-stage reads done before the next first stage writes
    }
}

// ---- inverse column pass + |.| combine + maximum -----------------------------------------------
// grid (tiles, cells), T = 8 columns per workgroup: a row of the tile is two 4-column groups, one
// 16-byte (fp16) or 32-byte (fp32) access per lane.  The second component's tile is fetched into
// registers while the first one is being transformed; the last stage turns its outputs into
// magnitudes in registers (no LDS round trip for the result).
template <int S, int T, int NCOMP, class ST>
__global__ __launch_bounds__((cols_threads<S, T>()), 3) void k_cols_inv_max_t(
    const float2 *__restrict__ tw, int L2, const ST *__restrict__ Bw, long L, float w0, float w1, int lo1, int hi1,
    int lo2, int hi2, Rec *__restrict__ recs, int rec_stride) {
    constexpr int NT = cols_threads<S, T>();
    constexpr int SP = tspan<S>();
    constexpr int QG = T / 4;        // 4-column groups per tile row
    constexpr int NI = S * QG / NT;  // (row, 4-column group) items per thread
    static_assert(S * QG % NT == 0, "tile items must divide evenly");
    static_assert(T == 4 || T == 8, "tile width");
    constexpr int RL = PlanInfo<S>::kLast, NSL = PlanInfo<S>::kNsLast;
    constexpr int TOTL = NSL * T, MBL = (TOTL + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];  // T * SP data + twiddle table
    float2 *tw_lds = lds + T * SP;
    load_twiddles<S, NT>(tw_lds, tw, threadIdx.x);
    __shared__ float s_v[NT / 64];
    __shared__ int s_l[NT / 64];
    const int tid = threadIdx.x;
    const int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int g = blockIdx.y;
    const int c0 = tile * T;
    const bool full_tile = c0 + T <= L2;  // L2 % 8 == 0 for every specialised length
    // lags >= hi_all are never searched: skip their magnitudes (the padded transform is ~1.6 N long)
    const int hi_all = hi1 > hi2 ? hi1 : hi2;
    C4 pre[NI];
    auto fetch = [&](int comp) {
        const ST *src = Bw + ((long)g * NCOMP + comp) * L;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + i * NT;
            const int r = it / QG, cq = (it % QG) * 4;
            if (full_tile) pre[i] = ld4(src, (long)r * L2 + c0 + cq);
        }
    };
    fetch(0);
    float mag[MBL][RL];
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + i * NT;
            const int r = it / QG, cq = (it % QG) * 4;
            const int pr = r + (r >> 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) lds[(cq + u) * SP + pr] = pre[i].v[u];
        }
        __syncthreads();
        if (comp + 1 < NCOMP) fetch(comp + 1);  // in flight during the transform
        const float w = comp == 0 ? w0 : w1;
        auto out = [&](int i, int q, int, int e, float2 v) {
            if ((long)e * L2 + c0 <= hi_all) {  // wave-uniform for the tail rows
                const float a = w * __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y);  // 1 ulp; feeds the sieve only
                mag[i][q] = comp == 0 ? a : mag[i][q] + a;
            }
        };
        TPlan<S>::template run<T, NT, +1>(lds, tw_lds, tid, LdsIO{}, out);
        if (comp + 1 < NCOMP) __syncthreads();  // last-stage reads done before the tile is overwritten
    }
    float bv = -1.f;
    int bl = -1;
#pragma unroll
    for (int i = 0; i < MBL; ++i) {
        const int b = tid + i * NT;
        if (TOTL % NT == 0 || b < TOTL) {
            const int j = b / NSL, bb = b - j * NSL;  // last stage: hi = 0, k = bb
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                const long lag = (long)(bb + q * NSL) * L2 + c0 + j;
                const bool in = full_tile && ((lag >= lo1 && lag <= hi1) || (lag >= lo2 && lag <= hi2));
                if (in) rec_better(bv, bl, mag[i][q], (int)lag);
            }
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

// =================================================================================================
// fp16-arithmetic sieve (packed v_pk_*_f16): same kernels with complex type h2 everywhere --
// registers, LDS, stage twiddles (per-stage [k][R] tables, pre-scaled so values stay near unit RMS).
// Half the VALU instructions and half the LDS bytes of the fp32 kernels; error ~2e-3 of the output
// RMS, covered by the wider refinement tolerance (the f64 refinement still makes every decision).
// =================================================================================================
__device__ __forceinline__ h2 ld_h(const __half2 *p, long i) {
    return *reinterpret_cast<const h2 *>(p + i);
}
template <int S, int NT>
__device__ __forceinline__ void load_half_table(h2 *__restrict__ dst, const h2 *__restrict__ src, int tid) {
    for (int i = tid; i < half_table_entries<S>(); i += NT) dst[i] = src[i];
}

struct RowsHArgs {
    const h2 *htab;
    TwiddleL twl;
    const __half2 *Xs;
    long L;
    int L1, G, bin0;
    const __half2 *Cs;
    __half2 *Bw;
    float in_scale;
    int GC;   // cells one workgroup walks through (same row k1 of GC consecutive Doppler bins)
    int NCH;  // = ceil(G / GC): workgroups per row
    // Optional cell list (GC must be 1): cell g is Doppler bin cell_bin[g] against the code spectra at
    // Cs + cell_cs[g] -- the B2a second-peak pass evaluates one (PRN, winning bin) cell per PRN in one launch.
    const int *cell_bin;
    const long *cell_cs;
};

// body of the fp16 row pass for virtual workgroup index vb (= 8*slot + xcd), thread tid < rows_threads<S>().
// A workgroup owns row k1 of up to GC cells of the group: the code-spectrum rows, the inter-pass
// twiddles and the LDS stage tables depend on (PRN, k1) only and are set up once; the spectrum row
// of the next cell is in flight while the current one is transformed.
template <int S, int NCOMP>
__device__ __forceinline__ void rows_inv_h_body(const RowsHArgs &A, int vb, int tid) {
    const h2 *__restrict__ htab = A.htab;
    const TwiddleL twl = A.twl;
    const __half2 *__restrict__ Xs = A.Xs;
    const long L = A.L;
    const int L1 = A.L1, G = A.G, bin0 = A.bin0;
    const __half2 *__restrict__ Cs = A.Cs;
    __half2 *__restrict__ Bw = A.Bw;
    const float in_scale = A.in_scale;
    constexpr int NT = rows_threads<S>();
    constexpr int NB1 = S / 16;
    constexpr int MB1 = (NB1 + NT - 1) / NT;
    constexpr int RL = PlanInfo<S>::kLast, NSL = PlanInfo<S>::kNsLast;
    constexpr int MBL = (NSL + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) h2 ldsh[];  // tspan<S>() data + stage tables
    __shared__ float2 s_a[MBL], s_b[RL];
    h2 *tab = ldsh + ((tspan<S>() + 3) & ~3);
    load_half_table<S, NT>(tab, htab, tid);
    const int xcd = vb & 7, m = vb >> 3;
    const int GC = A.GC, NCH = A.NCH;
    const int g0 = (m % NCH) * GC, k1 = (m / NCH) * 8 + xcd;
    const int g1 = g0 + GC < G ? g0 + GC : G;
    if (k1 >= L1) return;
    if (A.cell_cs) Cs += A.cell_cs[g0];
    if (tid < MBL) s_a[tid] = twl.get<+1>((uint32_t)((long)k1 * NT * tid));
    if (tid >= 64 && tid < 64 + RL) s_b[tid - 64] = twl.get<+1>((uint32_t)((long)k1 * NSL * (tid - 64)));
    const float2 wbase = twl.get<+1>((uint32_t)k1 * (uint32_t)tid);
    (void)in_scale;  // == 1: the unit-RMS scale is part of the stored spectrum (sX)
    h2 xn[MB1][16];  // spectrum row of the next cell (raw)
    auto fetch_x = [&](int g) {
        const int bin = A.cell_bin ? A.cell_bin[g] : bin0 + g;
        const __half2 *xr = Xs + (long)bin * L + (long)k1 * S;
#pragma unroll
        for (int i = 0; i < MB1; ++i) {
            const int bb = tid + i * NT;
            if (NB1 % NT == 0 || bb < NB1) {
#pragma unroll
                for (int q = 0; q < 16; ++q) xn[i][q] = ld_h(xr, bb + q * NB1);
            }
        }
    };
#ifndef BDS_ROWS_PREFETCH
#define BDS_ROWS_PREFETCH 1
#endif
#ifndef BDS_ROWS_OCC
#define BDS_ROWS_OCC 3
#endif
    if (BDS_ROWS_PREFETCH) fetch_x(g0);
    // code-spectrum rows of every component: all global loads of the workgroup are in flight
    // before the first transform starts
    h2 cv[NCOMP][MB1][16];
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
        const __half2 *cr = Cs + (long)comp * L + (long)k1 * S;
#pragma unroll
        for (int i = 0; i < MB1; ++i) {
            const int bb = tid + i * NT;
            if (NB1 % NT == 0 || bb < NB1) {
#pragma unroll
                for (int q = 0; q < 16; ++q) cv[comp][i][q] = ld_h(cr, bb + q * NB1);
            }
        }
    }
    __syncthreads();
    // inter-pass twiddle W_L^(-k1 e) of this thread's outputs: built in fp32 (base * step_i * step_q),
    // rounded to fp16 once per row and shared by the components
    h2 wo[MBL][RL];
#pragma unroll
    for (int i = 0; i < MBL; ++i) {
        const float2 wi = cmul(wbase, s_a[i]);
#pragma unroll
        for (int q = 0; q < RL; ++q) {
            const float2 w = cmul(wi, s_b[q]);
            wo[i][q] = h2{(_Float16)w.x, (_Float16)w.y};
        }
    }
    for (int g = g0; g < g1; ++g) {
        if (!BDS_ROWS_PREFETCH) fetch_x(g);
#pragma unroll
        for (int comp = 0; comp < NCOMP; ++comp) {
            h2 *dst = reinterpret_cast<h2 *>(Bw + ((long)g * NCOMP + comp) * L + (long)k1 * S);
            auto src = [&](int i, int q, int, int) { return cmul(xn[i][q], cv[comp][i][q]); };
            auto out = [&](int i, int q, int, int e, h2 v) { dst[e] = cmul(v, wo[i][q]); };
            // the last component's first stage is the last reader of xn: the next cell's row is
            // fetched into the same registers while stages 2.. and the stores run
            auto next = [&]() {
                if (BDS_ROWS_PREFETCH && comp == NCOMP - 1 && g + 1 < g1) fetch_x(g + 1);
            };
            TPlan<S>::template run_hook<1, NT, +1>(ldsh, (const h2 *)tab, tid, src, out, next);
            if (comp + 1 < NCOMP || g + 1 < g1) __syncthreads();
        }
    }
}

struct ColsHArgs {
    const h2 *htab;
    int L2;
    const __half2 *Bw;
    long L;
    float w0, w1;
    int lo1, hi1, lo2, hi2;
    Rec *recs;
    int rec_stride;  // = tiles per cell
    const int4 *cell_rng;  // optional per-cell (lo1, hi1, lo2, hi2), MASKED kernels only
};

// body of the fp16 column pass for tile index tb (of ntb = tiles per cell) of cell g, thread tid < cols_threads<S,T>()
// MASKED = false: the full search (lags 0..hi1, one range); true: the two ranges of the B2a second-peak pass.
template <int S, int T, int NCOMP, bool MASKED>
__device__ __forceinline__ void cols_inv_max_h_body(const ColsHArgs &A, int tb, int ntb, int g, int tid) {
    const h2 *__restrict__ htab = A.htab;
    const int L2 = A.L2;
    const __half2 *__restrict__ Bw = A.Bw;
    const long L = A.L;
    const float w0 = A.w0, w1 = A.w1;
    int lo1 = A.lo1, hi1 = A.hi1, lo2 = A.lo2, hi2 = A.hi2;
    if (MASKED && A.cell_rng) {
        const int4 r = A.cell_rng[g];
        lo1 = r.x, hi1 = r.y, lo2 = r.z, hi2 = r.w;
    }
    Rec *__restrict__ recs = A.recs;
    const int rec_stride = A.rec_stride;
    constexpr int NT = cols_threads<S, T>();
    constexpr int SP = tspan<S>();
    constexpr int QG = T / 4;
    constexpr int NI = S * QG / NT;  // (row, 4-column group) items per thread
    static_assert(S * QG % NT == 0, "tile items must divide evenly");
    constexpr int RL = PlanInfo<S>::kLast, NSL = PlanInfo<S>::kNsLast;
    constexpr int TOTL = NSL * T, MBL = (TOTL + NT - 1) / NT;
    static_assert(T == 4 || T == 8, "tile width");
    extern __shared__ __attribute__((aligned(16))) h2 ldsh[];  // T * SP data + stage tables
    h2 *tab = ldsh + ((T * SP + 3) & ~3);
    load_half_table<S, NT>(tab, htab, tid);
    __shared__ float s_v[NT / 64];
    __shared__ int s_l[NT / 64];
    const int tile = (int)xcd_remap((uint32_t)tb, (uint32_t)ntb);
    const int c0 = tile * T;
    const bool full_tile = c0 + T <= L2;
    const int hi_all = hi1 > hi2 ? hi1 : hi2;
    const int e_max = hi_all >= c0 ? (hi_all - c0) / L2 : -1;  // last output row that can hold a searched lag
    uint4 pre[NI];
    auto fetch = [&](int comp) {
        const __half2 *src = Bw + ((long)g * NCOMP + comp) * L;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + i * NT;
            const int r = it / QG, cq = (it % QG) * 4;
            if (full_tile) pre[i] = *reinterpret_cast<const uint4 *>(src + (long)r * L2 + c0 + cq);
        }
    };
    fetch(0);
    float mag[MBL][RL];
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + i * NT;
            const int r = it / QG, cq = (it % QG) * 4;
            const int pr = r + (r >> 4);
            union {
                uint4 u;
                h2 h[4];
            } t;
            t.u = pre[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) ldsh[(cq + u) * SP + pr] = t.h[u];
        }
        __syncthreads();
        if (comp + 1 < NCOMP) fetch(comp + 1);
        const float w = comp == 0 ? w0 : w1;
        // Output q of the last stage covers rows q*NSL .. q*NSL+NSL-1: a whole q beyond the searched
        // lags (the padded transform is ~1.6 N long) is skipped with a workgroup-uniform test.
        auto out = [&](int i, int q, int, int, h2 v) {
            if (q * NSL <= e_max) {
                // (v_dot2_f32_f16 for |v|^2 measured 1% slower than convert + fma: tools/exp_mag.sh)
                const float x = (float)v.x, y = (float)v.y;
                // raw v_sqrt_f32 (1 ulp): sqrtf() expands to ~12 instructions of denormal scaling and
                // Newton fix-up, a third of this kernel's scalar-rate VALU work; the value only feeds the sieve
                const float a = w * __builtin_amdgcn_sqrtf(x * x + y * y);
                mag[i][q] = comp == 0 ? a : mag[i][q] + a;
            }
        };
        TPlan<S>::template run<T, NT, +1>(ldsh, (const h2 *)tab, tid, LdsIO{}, out);
        if (comp + 1 < NCOMP) __syncthreads();
    }
    float bv = -1.f;
    int bl = -1;
#pragma unroll
    for (int i = 0; i < MBL; ++i) {
        const int b = tid + i * NT;
        if (TOTL % NT == 0 || b < TOTL) {
            const int j = b / NSL, bb = b - j * NSL;
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                if (q * NSL <= e_max) {
                    const int lag = (bb + q * NSL) * L2 + c0 + j;  // L < 2^31
                    bool in = full_tile && lag <= hi1;
                    if (MASKED) in = full_tile && ((lag >= lo1 && lag <= hi1) || (lag >= lo2 && lag <= hi2));
                    // ties inside one thread: whichever comes first -- every record within the
                    // tolerance band is re-evaluated in f64 anyway
                    if (in && mag[i][q] > bv) {
                        bv = mag[i][q];
                        bl = lag;
                    }
                }
            }
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

template <int S, int NCOMP>
__global__ __launch_bounds__(rows_threads<S>(), BDS_ROWS_OCC) void k_rows_inv_h(RowsHArgs A) {
    rows_inv_h_body<S, NCOMP>(A, (int)blockIdx.x, (int)threadIdx.x);
}

template <int S, int T, int NCOMP, bool MASKED>
__global__ __launch_bounds__((cols_threads<S, T>()), 4) void k_cols_inv_max_h(ColsHArgs A) {
    cols_inv_max_h_body<S, T, NCOMP, MASKED>(A, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)threadIdx.x);
}

// ---- fused launch: row pass of cell group k+1 beside the column pass of group k -----------------
// The row pass is HBM-bound (it writes the inter-pass buffer), the column pass VALU-bound; as
// separate launches they run back to back.  Here one grid carries both kinds of workgroup,
// interleaved in proportion (Bresenham over 8-wide "slots" so that workgroup b still lands on XCD
// b % 8 with the slot's partners), and the CU's wave slots hold a mix of the two.  The two groups
// use different halves of the inter-pass buffer.  nr / nc = row / column slots (workgroups / 8);
// either may be 0 (first / last launch of the chain).
template <int S2, int S1, int T, int NCOMP>
__global__ __launch_bounds__(rows_threads<S2>(), BDS_ROWS_OCC) void k_search_fused_h(RowsHArgs RA, ColsHArgs CA, int nr, int nc,
                                                                        int ntiles) {
    static_assert(rows_threads<S2>() >= cols_threads<S1, T>(), "block size is the row pass's");
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const long tot = (long)nr + nc;
    const int sr = (int)(((long)slot * nr) / tot);          // row slots before this one
    const bool is_row = (int)(((long)(slot + 1) * nr) / tot) > sr;
    const int tid = threadIdx.x;
    if (is_row) {
        rows_inv_h_body<S2, NCOMP>(RA, sr * 8 + xcd, tid);
    } else {
        if (tid >= cols_threads<S1, T>()) return;  // surplus wave of the wider block
        const int v = (slot - sr) * 8 + xcd;       // column workgroup index: cell-major
        cols_inv_max_h_body<S1, T, NCOMP, false>(CA, v % ntiles, ntiles, v / ntiles, tid);
    }
}

}  // namespace bds