// fp32-arithmetic search kernels for the specialised transform plans (gfx950): the default search path.
//
// Same two passes and the same HBM layout as the kernels in bds_acq_fast.h:
//   row pass    : spectrum product X_b .* conj(C_p) + inverse rows (length L2) + inter-pass twiddle
//   column pass : inverse columns (length L1) + w_d|y_d| + w_p|y_p| + maximum per tile
// All arithmetic is fp32 (v_fma/v_add/v_mul_f32, the full-rate VALU class on gfx950: tools/probe/valu_rate.hip
// measures 2 cycles per wave-instruction for those against 4 for every v_pk_*_f16, so packed fp16 buys no
// issue time).  ST is the HBM storage type of the spectra and of the inter-pass buffer:
//   __half2 : fp16 complex (default) -- products are formed with v_dot2_f32_f16 (exact fp16 x fp16
//             products, fp32 accumulation) straight from the packed operands, results are rounded to
//             nearest with v_cvt_pk_f16_f32
//   float2  : fp32 complex
// What the row pass takes from the fp16 kernels' structure: one workgroup owns spectrum row k1 for up to GC
// cells (code-spectrum rows, inter-pass twiddles and the LDS twiddle table are set up once; the next cell's
// spectrum row is in flight while the current one is transformed), optional cell lists.
// The column pass reports per tile the maximum AND every other lag whose value is within `keep` of it
// (overflow list), so that no candidate of the f64 refinement can hide behind a larger neighbour of its tile.
#pragma once

#include "bds_acq_fast.h"

// Timing experiments (tools/exp/exp_parts.sh; results are INVALID with any of these defined):
//   BDS_EXP_NOBARRIER  __syncthreads() of the two search kernels compiled out
//   BDS_EXP_ROWS_NOSTORE / BDS_EXP_COLS_NOLOAD  no inter-pass buffer traffic
//   BDS_EXP_ROWS_OCC   launch bound (waves per SIMD) of the row pass
#ifdef BDS_EXP_NOBARRIER
#define BDS_SYNC() __builtin_amdgcn_s_waitcnt(0)
#else
#define BDS_SYNC() __syncthreads()
#endif
// Row pass at 2 waves per SIMD: it needs ~216 VGPRs (two packed code-spectrum rows, the inter-pass twiddles and a
// radix-16 butterfly with its twiddles live at once); squeezed into the 168 of a third wave it spills ~48 of them
// to scratch inside the cell loop, which doubles the kernel's HBM reads (measured 2.4 ms vs 2.0 ms per 201 cells).
#ifndef BDS_EXP_ROWS_OCC
#define BDS_EXP_ROWS_OCC 2
#endif

// BDS_EXP_PHASES (tools/exp/phases.sh; a timing build, never the product): the wave-private search kernels stamp the shader clock
// (s_memtime) at their phase boundaries, with explicit waits so that memory / LDS / barrier waits are told apart from issue
// time, and add the intervals to g_phase[]; read and cleared by bds_debug_phases().
#ifdef BDS_EXP_PHASES
__device__ unsigned long long g_phase[128];
#define PH_DECL(n)                       \
    unsigned long long ph_acc[n] = {};   \
    unsigned long long ph_t = __builtin_readcyclecounter()
#define PH_MARK(i)                                                   \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        const unsigned long long t_ = __builtin_readcyclecounter();  \
        ph_acc[i] += t_ - ph_t;                                      \
        ph_t = t_;                                                   \
        __builtin_amdgcn_sched_barrier(0);                           \
    } while (0)
#define PH_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PH_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PH_FLUSH(base, n)                                                            \
    do {                                                                             \
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 127) == 5) { /* sampled: same-line atomics run at ~10 M/s */ \
            for (int i_ = 0; i_ < (n); ++i_) atomicAdd(&g_phase[(base) + i_], ph_acc[i_]); \
            atomicAdd(&g_phase[(base) + (n)], 1ull);                                 \
        }                                                                            \
    } while (0)
#else
#define PH_DECL(n)
#define PH_MARK(i)
#define PH_WAIT_VM()
#define PH_WAIT_LGKM()
#define PH_FLUSH(base, n)
#endif

namespace bds {

// Per-stage [q][k] twiddle tables in LDS for the inverse fp32 transforms (bds_fft_t.h tstage TAB) instead of the W_S
// table + products: -44 vector instructions per radix-16 butterfly and twiddled stage for 8-35 KB of LDS and 11 more
// LDS reads.  Measured on cfg3: column pass 2.32 vs 2.38 ms, row pass 2.12 vs 2.10 ms (its LDS pipe is as busy as its
// vector unit) -- so the column pass takes the tables and the row pass keeps the products.
#ifndef BDS_F32_TAB_COLS
#define BDS_F32_TAB_COLS 1
#endif
#ifndef BDS_F32_TAB_ROWS
#define BDS_F32_TAB_ROWS 0
#endif
static constexpr bool kF32TabRows = BDS_F32_TAB_ROWS != 0;
// column pass: only the 768-point plan (cfg3) -- the 256-point kernel, whose twiddled stage is also its register-heavy
// last stage, spills 25 VGPRs with the tables and runs 2x slower (cfg2: 2.95 vs 1.43 ms)
template <int S>
__host__ __device__ constexpr bool f32_tab_cols() {
    return BDS_F32_TAB_COLS != 0 && S == 768;
}
// float2 entries of the twiddle area behind the data in LDS
template <int S, bool TAB>
__host__ __device__ constexpr int f32_tw_span() {
    return TAB ? half_table_entries<S>() : lds_span(twiddle_entries<S>());
}


typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ h2 as_h2(uint32_t u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ uint32_t as_u32(h2 v) { return __builtin_bit_cast(uint32_t, v); }
// fp16 complex -> fp32 complex
__device__ __forceinline__ float2 h2_to_f2(uint32_t u) {
    const f2v f = __builtin_convertvector(as_h2(u), f2v);
    return make_float2(f.x, f.y);
}
// fp32 complex -> fp16 complex, round to nearest even (v_cvt_pk_f16_f32 on gfx950)
__device__ __forceinline__ uint32_t f2_to_h2(float2 v) {
    const f2v f = {v.x, v.y};
    return as_u32(__builtin_convertvector(f, h2));
}

// (xr + j xi)(cr + j ci) from packed fp16 operands: two v_dot2_f32_f16 (products exact, sums in fp32).
// xswp = (xi, xr) is prepared once per spectrum element and shared by the components; (xr, -xi) is one v_xor.
// (Compiler builtin, not inline asm: the dot instructions have issue hazards on gfx950 that only the compiler's
// hazard recognizer pads -- an asm v_dot2 with a neg_hi modifier returned wrong sums on the 2048/4096 plans.)
__device__ __forceinline__ float2 cmul_h(uint32_t x, uint32_t xswp, uint32_t c) {
    return make_float2(__builtin_amdgcn_fdot2(as_h2(x ^ 0x80000000u), as_h2(c), 0.f, false),
                       __builtin_amdgcn_fdot2(as_h2(xswp), as_h2(c), 0.f, false));
}

struct RowsFArgs {
    const float2 *tw;   // W_S table of the row transform (fp32, exp(-j..))
    TwiddleL twl;
    const void *Xs;
    long L;
    int L1, G, bin0;
    const void *Cs;
    void *Bw;
    float out_scale;    // power-of-two storage scale of the inter-pass buffer (1 for fp32 storage)
    int GC;             // cells one workgroup walks through (same row k1 of GC consecutive cells)
    int NCH;            // = ceil(G / GC): workgroups per row
    const int *cell_bin;   // optional cell list: Doppler bin of cell g ...
    const long *cell_cs;   // ... and element offset of its code spectra from Cs
    int nvb;               // virtual workgroups (= L1 * NCH); a launch with fewer workgroups strides over them
    unsigned long long *clk;  // optional (BDS_ACQ_CLOCKPROBE): [0], [1] += shader-clock / reference-clock ticks of sampled workgroups
    int ilv;                  // k_rows_wave_f with two components: inter-pass buffer laid out [cell][element][component]
};

// Engine clock under the real load (BDS_ACQ_CLOCKPROBE=1): every (mask + 1)-th workgroup times its own life with the shader
// clock (s_memtime) and the constant reference clock (s_memrealtime); the host turns the two sums into GHz
// (bds_timing::shader_clock_GHz).  All scalar; off (null pointer) it costs one scalar compare.
struct ClockProbe {
    unsigned long long *acc;
    long long c0 = 0, r0 = 0;
#ifdef BDS_EXP_NOCLOCK
    __device__ __forceinline__ ClockProbe(unsigned long long *, unsigned) : acc(nullptr) {}
#else
    __device__ __forceinline__ ClockProbe(unsigned long long *p, unsigned mask) : acc((p && (blockIdx.x & mask) == 0) ? p : nullptr) {
        if (acc) c0 = clock64(), r0 = wall_clock64();
    }
#endif
    __device__ __forceinline__ void finish(int tid) const {
        if (acc) {
            const long long c1 = clock64(), r1 = wall_clock64();
            if (tid == 0) {
                atomicAdd(acc, (unsigned long long)(c1 - c0));
                atomicAdd(acc + 1, (unsigned long long)(r1 - r0));
            }
        }
    }
};

// ---- inverse row pass ------------------------------------------------------------------------------
template <int S, int NCOMP, class ST>
__device__ __forceinline__ void rows_inv_f_body(const RowsFArgs &A, int vb, int tid) {
    constexpr bool HS = std::is_same<ST, __half2>::value;
    constexpr int NT = rows_threads<S>();
    constexpr int NB1 = S / 16;
    constexpr int MB1 = (NB1 + NT - 1) / NT;
    constexpr int RL = PlanInfo<S>::kLast, NSL = PlanInfo<S>::kNsLast;
    constexpr int MBL = (NSL + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float2 ldsf[];  // tspan<S>() data + twiddle table
    __shared__ float2 s_a[MBL], s_b[RL];
    float2 *tw_lds = ldsf + tspan<S>();
    if constexpr (kF32TabRows)
        load_stage_tables<S, NT>(tw_lds, A.tw, tid);
    else
        load_twiddles<S, NT>(tw_lds, A.tw, tid);
    const long L = A.L;
    const int xcd = vb & 7, m = vb >> 3;
    const int GC = A.GC, NCH = A.NCH;
    const int g0 = (m % NCH) * GC, k1 = (m / NCH) * 8 + xcd;
    const int g1 = g0 + GC < A.G ? g0 + GC : A.G;
    const ST *Cs = (const ST *)A.Cs;
    if (A.cell_cs) Cs += A.cell_cs[g0];
    if (tid < MBL) s_a[tid] = A.twl.get<+1>((uint32_t)((long)k1 * NT * tid));
    if (tid >= 64 && tid < 64 + RL) s_b[tid - 64] = A.twl.get<+1>((uint32_t)((long)k1 * NSL * (tid - 64)));
    const float2 wbase = A.twl.get<+1>((uint32_t)k1 * (uint32_t)tid);

    // spectrum row of the next cell, raw as stored
    typename std::conditional<HS, uint32_t, float2>::type xn[MB1][16];
    auto fetch_x = [&](int g) {
        const int bin = A.cell_bin ? A.cell_bin[g] : A.bin0 + g;
        const ST *xr = (const ST *)A.Xs + (long)bin * L + (long)k1 * S;
#pragma unroll
        for (int i = 0; i < MB1; ++i) {
            const int bb = tid + i * NT;
            if (NB1 % NT == 0 || bb < NB1) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if constexpr (HS)
                        xn[i][q] = *reinterpret_cast<const uint32_t *>(xr + bb + q * NB1);
                    else
                        xn[i][q] = xr[bb + q * NB1];
                }
            }
        }
    };
    fetch_x(g0);
    // fp16 storage: the code-spectrum rows of every component stay in registers (packed) for all cells
    uint32_t cv[HS ? NCOMP : 1][MB1][16];
    if constexpr (HS) {
#pragma unroll
        for (int comp = 0; comp < NCOMP; ++comp) {
            const ST *cr = Cs + (long)comp * L + (long)k1 * S;
#pragma unroll
            for (int i = 0; i < MB1; ++i) {
                const int bb = tid + i * NT;
                if (NB1 % NT == 0 || bb < NB1) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) cv[comp][i][q] = *reinterpret_cast<const uint32_t *>(cr + bb + q * NB1);
                }
            }
        }
    }
    __syncthreads();  // twiddle table, s_a, s_b
    // inter-pass twiddle W_L^(-k1 e) of this thread's outputs e = tid + i NT + q NSL
    float2 wo[MBL][RL];
#pragma unroll
    for (int i = 0; i < MBL; ++i) {
        float2 wi = cmul(wbase, s_a[i]);
        wi.x *= A.out_scale;
        wi.y *= A.out_scale;
#pragma unroll
        for (int q = 0; q < RL; ++q) wo[i][q] = cmul(wi, s_b[q]);
    }
    for (int g = g0; g < g1; ++g) {
        // (xi, xr) of the current cell's spectrum row, shared by the components (fp16 storage)
        uint32_t xs[HS ? MB1 : 1][16];
        if constexpr (HS) {
#pragma unroll
            for (int i = 0; i < MB1; ++i) {
#pragma unroll
                for (int q = 0; q < 16; ++q) xs[i][q] = __builtin_amdgcn_alignbit(xn[i][q], xn[i][q], 16);
            }
        }
#pragma unroll
        for (int comp = 0; comp < NCOMP; ++comp) {
            ST *dst = (ST *)A.Bw + ((long)g * NCOMP + comp) * L + (long)k1 * S;
            const ST *cr = Cs + (long)comp * L + (long)k1 * S;
            auto src = [&](int i, int q, int, int e) {
                if constexpr (HS) {
                    (void)e;
                    return cmul_h(xn[i][q], xs[i][q], cv[comp][i][q]);
                } else {
                    return cmul(xn[i][q], cr[e]);
                }
            };
            auto out = [&](int i, int q, int, int e, float2 v) {
                const float2 t = cmul(v, wo[i][q]);
#ifdef BDS_EXP_ROWS_NOSTORE
                if (t.x == 1.2345f)
#endif
                if constexpr (HS)
                    *reinterpret_cast<uint32_t *>(dst + e) = f2_to_h2(t);
                else
                    dst[e] = t;
            };
            // the last component's first stage is the last reader of xn: the next cell's row is fetched into
            // the same registers while stages 2.. and the stores run
            auto next = [&]() {
                if (comp == NCOMP - 1 && g + 1 < g1) fetch_x(g + 1);
            };
            TPlan<S>::template run_hook<1, NT, +1, kF32TabRows>(ldsf, (const float2 *)tw_lds, tid, src, out, next);
            if (comp + 1 < NCOMP || g + 1 < g1) BDS_SYNC();  // last-stage reads precede the next first-stage writes
        }
    }
}

// ---- inverse column pass + |.| combine + maximum ---------------------------------------------------
struct Extra {
    float v;
    int lag;   // 0-based
    int cell;  // cell index within the run (PRN index * D + bin, or the second-peak pass's PRN index)
};

struct ColsFArgs {
    const float2 *tw;   // W_S table of the column transform
    int L2;
    const void *Bw;
    long L;
    float w0, w1;
    int lo1, hi1, lo2, hi2;
    Rec *recs;
    int rec_stride;         // tiles per cell
    const int4 *cell_rng;   // optional per-cell (lo1, hi1, lo2, hi2), MASKED kernels only
    Extra *extra;           // overflow list: lags within `keep` of their tile's maximum (other than the record)
    int *extra_count;
    int extra_cap;
    int cell0;              // run-wide index of cell 0 of this launch
    float keep;             // 1 - tolerance of the sieve
};

// maximum over the 64 lanes, wave-uniform result (DPP inside the 16-lane rows, then one read per row)
__device__ __forceinline__ float wave_max_f32(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}));  // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

template <int S, int T, int NCOMP, bool MASKED, class ST>
__device__ __forceinline__ void cols_inv_max_f_body(const ColsFArgs &A, int tb, int ntb, int g, int tid) {
    constexpr bool HS = std::is_same<ST, __half2>::value;
    const int L2 = A.L2;
    const long L = A.L;
    int lo1 = A.lo1, hi1 = A.hi1, lo2 = A.lo2, hi2 = A.hi2;
    if (MASKED && A.cell_rng) {
        const int4 r = A.cell_rng[g];
        lo1 = r.x, hi1 = r.y, lo2 = r.z, hi2 = r.w;
    }
    constexpr int NT = cols_threads<S, T>();
    constexpr int NW = NT / 64;
    constexpr int SP = tspan<S>();
    constexpr int QG = T / 4;
    constexpr int NI = S * QG / NT;  // (row, 4-column group) items per thread
    static_assert(S * QG % NT == 0, "tile items must divide evenly");
    static_assert(T == 4 || T == 8, "tile width");
    constexpr int RL = PlanInfo<S>::kLast, NSL = PlanInfo<S>::kNsLast;
    constexpr int TOTL = NSL * T, MBL = (TOTL + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float2 ldsf[];  // T * SP data + twiddle table
    float2 *tw_lds = ldsf + T * SP;
    if constexpr (f32_tab_cols<S>())
        load_stage_tables<S, NT>(tw_lds, A.tw, tid);
    else
        load_twiddles<S, NT>(tw_lds, A.tw, tid);
    __shared__ float s_v[NW];
    __shared__ int s_cnt;
    if (tid == 0) s_cnt = 0;
    const int tile = (int)xcd_remap((uint32_t)tb, (uint32_t)ntb);
    const int c0 = tile * T;
    const bool full_tile = c0 + T <= L2;  // L2 % 8 == 0 for every specialised length
    const int hi_all = hi1 > hi2 ? hi1 : hi2;
    const int e_max = hi_all >= c0 ? (hi_all - c0) / L2 : -1;  // last output row that can hold a searched lag
    typename std::conditional<HS, uint4, C4>::type pre[NI];
    auto fetch = [&](int comp) {
        const ST *src = (const ST *)A.Bw + ((long)g * NCOMP + comp) * L;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + i * NT;
            const int r = it / QG, cq = (it % QG) * 4;
#ifdef BDS_EXP_COLS_NOLOAD
            if (full_tile && A.w0 == 1.2345f) {
#else
            if (full_tile) {
#endif
                if constexpr (HS)
                    pre[i] = *reinterpret_cast<const uint4 *>(src + (long)r * L2 + c0 + cq);
                else
                    pre[i] = ld4(src, (long)r * L2 + c0 + cq);
            }
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
            if constexpr (HS) {
                const uint32_t u[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) ldsf[(cq + k) * SP + pr] = h2_to_f2(u[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) ldsf[(cq + k) * SP + pr] = pre[i].v[k];
            }
        }
        BDS_SYNC();
        if (comp + 1 < NCOMP) fetch(comp + 1);  // in flight during the transform
        const float w = comp == 0 ? A.w0 : A.w1;
        // Output q of the last stage covers rows q*NSL .. q*NSL+NSL-1: a whole q beyond the searched lags
        // (the padded transform is ~1.6 N long) is skipped with a workgroup-uniform test.
        auto out = [&](int i, int q, int, int, float2 v) {
            if (q * NSL <= e_max) {
                // raw v_sqrt_f32 (1 ulp): the value only feeds the sieve
                const float a = w * __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y);
                mag[i][q] = comp == 0 ? a : mag[i][q] + a;
            }
        };
        TPlan<S>::template run<T, NT, +1, f32_tab_cols<S>()>(ldsf, (const float2 *)tw_lds, tid, LdsIO{}, out);
        if (comp + 1 < NCOMP) BDS_SYNC();  // last-stage reads done before the tile is overwritten
    }
    // lag of output (i, q): (bb + q NSL) L2 + c0 + j with b = tid + i NT, j = b / NSL, bb = b % NSL
    auto lag_of = [&](int i, int q) {
        const int b = tid + i * NT;
        const int j = b / NSL, bb = b - j * NSL;
        return (bb + q * NSL) * L2 + c0 + j;  // L < 2^31
    };
    auto valid = [&](int i, int q) -> bool {
        const int b = tid + i * NT;
        if (!(TOTL % NT == 0 || b < TOTL) || !full_tile || q * NSL > e_max) return false;
        if (!MASKED && (q + 1) * NSL <= e_max) return true;  // every row of this q lies below the last searched row
        const int lag = lag_of(i, q);
        if (MASKED) return (lag >= lo1 && lag <= hi1) || (lag >= lo2 && lag <= hi2);
        return lag <= hi1;
    };
    float mx = -1.f;
#pragma unroll
    for (int i = 0; i < MBL; ++i) {
#pragma unroll
        for (int q = 0; q < RL; ++q)
            if (valid(i, q)) mx = fmaxf(mx, mag[i][q]);
    }
    const float wm = wave_max_f32(mx);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) s_v[wave] = wm;
    __syncthreads();
    float Mt = s_v[0];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) Mt = fmaxf(Mt, s_v[w2]);
    Rec *rec = A.recs + (long)g * A.rec_stride + tile;
    if (!(Mt > 0.f)) {  // nothing searched in this tile, or an all-zero surface: first searched lag, like max()
        if (tid == 0) {
            Rec rr;
            rr.v = Mt;
            rr.lag = (!MASKED && full_tile && Mt == 0.f && c0 <= hi1) ? c0 : -1;
            *rec = rr;
        }
        return;
    }
    const float thr = Mt * A.keep;
#ifndef BDS_NO_SLOWPATH
    if (mx >= thr) {  // rare: this thread holds the maximum or a value within the sieve tolerance of it
#pragma unroll
        for (int i = 0; i < MBL; ++i) {
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                if (valid(i, q) && mag[i][q] >= thr) {
                    const int k = atomicAdd(&s_cnt, 1);
                    const int lag = lag_of(i, q);
                    if (k == 0) {  // the tile's record: its maximum with one lag of the tolerance band
                        Rec rr;
                        rr.v = Mt;
                        rr.lag = lag;
                        *rec = rr;
                    } else {
                        const int idx = atomicAdd(A.extra_count, 1);
                        if (idx < A.extra_cap) {
                            Extra ex;
                            ex.v = mag[i][q];
                            ex.lag = lag;
                            ex.cell = A.cell0 + g;
                            A.extra[idx] = ex;
                        }
                    }
                }
            }
        }
    }
#endif
}

template <int S, int NCOMP, class ST>
__global__ __launch_bounds__(rows_threads<S>(), BDS_EXP_ROWS_OCC) void k_rows_inv_f(RowsFArgs A) {
    // (a grid smaller than nvb makes the workgroups persistent: with the column pass of the previous group on a
    //  second stream they then share every CU with it instead of queueing in front of it)
    for (int vb = (int)blockIdx.x; vb < A.nvb; vb += (int)gridDim.x) {
        rows_inv_f_body<S, NCOMP, ST>(A, vb, (int)threadIdx.x);
        if (vb + (int)gridDim.x < A.nvb) BDS_SYNC();  // the next item rewrites the LDS tables
    }
}

#ifndef BDS_COLS_MINW
#define BDS_COLS_MINW 4
#endif
template <int S, int T, int NCOMP, bool MASKED, class ST>
__global__ __launch_bounds__((cols_threads<S, T>()), BDS_COLS_MINW) void k_cols_inv_max_f(ColsFArgs A) {
    cols_inv_max_f_body<S, T, NCOMP, MASKED, ST>(A, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)threadIdx.x);
}

}  // namespace bds
