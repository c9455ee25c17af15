// Compile-time specialised LDS transform stages for the hot plans (gfx950).
//
// Same algorithm and LDS layout as the run-time engine in bds_fft.h (mixed-radix Stockham
// autosort, logical element i at physical i + (i >> 4)), but the transform length S, the
// batch T, the workgroup size NT and the radix list are template parameters: butterfly ->
// thread assignment, LDS strides and the autosort scatter all become immediates (the generic
// engine spends ~3/4 of its instructions on that index arithmetic, not on the butterflies).
// Requirements, checked with static_assert: the first radix is 16 and S/R is a multiple of 16
// for every stage, i.e. S = 256*m; later stages then have NS a multiple of 16 and both the
// gather and the scatter of a butterfly are "base + q * constant".
#pragma once

#include <type_traits>

#include "bds_fft.h"
#include "bds_fft_fma.h"

#ifdef BDS_EXP_NOBARRIER
#define BDS_TSYNC() __builtin_amdgcn_s_waitcnt(0)
#else
#define BDS_TSYNC() __syncthreads()
#endif

namespace bds {

template <int S>
__host__ __device__ constexpr int tspan() { return lds_span(S) + 4; }  // per-transform LDS stride (elements)

// Tag: the stage reads its inputs from / writes its outputs to the LDS transform buffer.
struct LdsIO {};

// scale folded into the fp16 stage-twiddle tables so that values neither grow nor vanish:
// a radix-R stage multiplies the RMS of noise-like data by sqrt(R)
__host__ __device__ constexpr float stage_scale(int R) { return R == 16 ? 0.25f : (R == 8 || R == 4) ? 0.5f : 1.0f; }

// One radix-R stage on complex type C (float2: fp32; h2: packed fp16).  Src / Dst are LdsIO or
// functors that replace the LDS side:
//   src(i, q, j, e)  -> C : input element e (logical index) of transform j (i = butterfly slot, q = input)
//   dst(i, q, j, e, v)    : output element e of transform j (i = butterfly slot, q = output)
// A functor side needs no barrier of its own; the caller orders it against other LDS traffic.
// Twiddles: fp32 -- tw is the LDS copy of the W_S table (physical layout), the power-of-two
// multiples are fetched and the rest built as products; fp16 -- tw is the LDS copy of the
// per-stage tables [R][NS] (q-major) at offset TWOFF, already in the transform direction and pre-scaled.
// JFAST: butterfly slot b -> (transform j = b % T, butterfly bb = b / T) instead of (j = b / NB, bb = b % NB):
// adjacent lanes then work on ADJACENT transforms (columns), which is what a first stage fed straight from a
// row-major global tile wants (T must be a power of two).
template <class C, int S, int T, int NT, int DIR, int NS, int R, int TWOFF, class Src, class Dst, bool JFAST = false, bool TAB = false>
__device__ __forceinline__ void tstage(C *__restrict__ buf, const C *__restrict__ tw, int tid, Src src, Dst dst) {
    constexpr int NB = S / R;
    constexpr int TOTAL = NB * T;
    constexpr int MB = (TOTAL + NT - 1) / NT;
    constexpr bool FULL = (TOTAL % NT) == 0;
    constexpr int SP = tspan<S>();
    constexpr int RSTR = NB + NB / 16;  // physical stride between the R inputs of a butterfly
    constexpr int WSTR = NS + NS / 16;  // physical stride between its outputs (NS >= 16)
    constexpr int TWS = S / (NS * R);   // stride into the W_S table
    constexpr bool SRC_LDS = std::is_same<Src, LdsIO>::value;
    constexpr bool DST_LDS = std::is_same<Dst, LdsIO>::value;
    constexpr bool HALF = std::is_same<C, h2>::value;
#ifdef BDS_EXP_PLAIN_TSTAGE
    constexpr bool kFma = false;
#else
    constexpr bool kFma = std::is_same<C, float2>::value && (R == 16 || R == 8);
#endif
    static_assert(NB % 16 == 0, "S/R must be a multiple of 16");
    static_assert(NS == 1 || NS % 16 == 0, "later stages need NS % 16 == 0");
    static_assert(NS > 1 || R == 16, "the first stage must be radix 16");
    C v[MB][R];
    int jj[MB], j0v[MB], kidx[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int b = tid + i * NT;
        if (FULL || b < TOTAL) {
            const int j = JFAST ? b % T : b / NB, bb = JFAST ? b / T : b - (b / NB) * NB;
            if constexpr (SRC_LDS) {
                const C *sp = buf + j * SP + bb + (bb >> 4);
#pragma unroll
                for (int q = 0; q < R; ++q) v[i][q] = sp[q * RSTR];
            } else {
#pragma unroll
                for (int q = 0; q < R; ++q) v[i][q] = src(i, q, j, bb + q * NB);
            }
            jj[i] = j;
            if (NS == 1) {
                j0v[i] = bb * R;
                kidx[i] = 0;
            } else {
                const int hi = bb / NS, k = bb - hi * NS;
                j0v[i] = hi * (NS * R) + k;
                kidx[i] = k;
            }
        }
    }
    if constexpr (SRC_LDS && DST_LDS) BDS_TSYNC();  // every read of this stage precedes its writes
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int b = tid + i * NT;
        if (FULL || b < TOTAL) {
            [[maybe_unused]] C wq[R];  // stage twiddles of this butterfly (kFma)
            if (NS > 1) {
                if constexpr (HALF) {
                    const C *tk = tw + TWOFF + kidx[i];  // table layout [q][k]: lanes (consecutive k) hit consecutive banks
#pragma unroll
                    for (int q = 1; q < R; ++q) v[i][q] = cmul(v[i][q], tk[q * NS]);
                    v[i][0] = cscale(v[i][0], stage_scale(R));
                } else if constexpr (TAB) {
                    // fp32 with per-stage tables (stage_tables_f32 on the host: [q][k], already in the transform
                    // direction): 15 LDS reads instead of 4 reads + 11 complex products per radix-16 butterfly
                    const C *tk = tw + TWOFF + kidx[i];
                    if constexpr (kFma) {
#pragma unroll
                        for (int q = 1; q < R; ++q) wq[q] = tk[q * NS];
                    } else {
#pragma unroll
                        for (int q = 1; q < R; ++q) v[i][q] = cmul(v[i][q], tk[q * NS]);
                    }
                } else {
                    const int kt = kidx[i] * TWS;
                    if constexpr (R == 16 || R == 8) {
                        C w[R];
                        w[1] = tw[lds_phys(kt)];  // twiddle table is LDS resident, padded like the data
                        w[2] = tw[lds_phys(2 * kt)];
                        w[4] = tw[lds_phys(4 * kt)];
                        if constexpr (R == 16) w[8] = tw[lds_phys(8 * kt)];
                        if (DIR > 0) {
                            w[1].y = -w[1].y;
                            w[2].y = -w[2].y;
                            w[4].y = -w[4].y;
                            if constexpr (R == 16) w[8].y = -w[8].y;
                        }
                        w[3] = cmul(w[1], w[2]);
                        w[5] = cmul(w[1], w[4]);
                        w[6] = cmul(w[2], w[4]);
                        w[7] = cmul(w[3], w[4]);
                        if constexpr (R == 16) {
#pragma unroll
                            for (int q = 1; q < 8; ++q) w[8 + q] = cmul(w[q], w[8]);
                        }
                        if constexpr (kFma) {
#pragma unroll
                            for (int q = 1; q < R; ++q) wq[q] = w[q];
                        } else {
#pragma unroll
                            for (int q = 1; q < R; ++q) v[i][q] = cmul(v[i][q], w[q]);
                        }
                    } else {
#pragma unroll
                        for (int q = 1; q < R; ++q) {
                            C w = tw[lds_phys(q * kt)];
                            if (DIR > 0) w.y = -w.y;
                            v[i][q] = cmul(v[i][q], w);
                        }
                    }
                }
            }
            // fp32, radix 16 / 8: the stage twiddles go into the butterfly's first layer (bds_fft_fma.h)
            if constexpr (kFma) {
                if constexpr (R == 16) {
                    if (NS > 1) bfly16_fma<DIR, true>(v[i], wq);
                    else bfly16_fma<DIR, false>(v[i], nullptr);
                } else {
                    if (NS > 1) bfly8_fma<DIR, true>(v[i], wq);
                    else bfly8_fma<DIR, false>(v[i], nullptr);
                }
            } else {
                Butterfly<R, DIR>::run(v[i]);
            }
            if constexpr (DST_LDS) {
                // phys(j0 + q*NS): NS == 1 -> 17*bb + q ; NS % 16 == 0 -> phys(j0) + q*WSTR
                C *dp = buf + jj[i] * SP + j0v[i] + (j0v[i] >> 4);
#pragma unroll
                for (int q = 0; q < R; ++q) dp[NS == 1 ? q : q * WSTR] = v[i][q];
            } else {
#pragma unroll
                for (int q = 0; q < R; ++q) dst(i, q, jj[i], j0v[i] + q * NS, v[i][q]);
            }
        }
    }
    if constexpr (DST_LDS) BDS_TSYNC();
}

// All stages of a plan; the first stage takes Src, the last one Dst, everything between is LDS.
// hook() runs once, right after the first stage (its inputs are consumed: the caller may start
// overwriting them, e.g. with the loads of the next transform's operands).
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <class C, int S, int T, int NT, int DIR, int NS, int TWOFF, bool JFAST, bool TAB, class Src, class Dst, class Hook, int R, int... REST>
__device__ __forceinline__ void tfft_run(C *__restrict__ buf, const C *__restrict__ tw, int tid, Src src, Dst dst, Hook hook) {
    if constexpr (sizeof...(REST) == 0) {
        tstage<C, S, T, NT, DIR, NS, R, TWOFF, Src, Dst, JFAST, TAB>(buf, tw, tid, src, dst);
        hook();
    } else {
        tstage<C, S, T, NT, DIR, NS, R, TWOFF, Src, LdsIO, JFAST, TAB>(buf, tw, tid, src, LdsIO{});
        hook();
        tfft_run<C, S, T, NT, DIR, NS * R, TWOFF + (NS > 1 ? NS * R : 0), false, TAB, LdsIO, Dst, NoHook, REST...>(buf, tw, tid, LdsIO{}, dst, NoHook{});
    }
}

// Entries of the W_S table a length-S plan can touch (defined after the plans: the largest index any stage forms).
template <int S>
__host__ __device__ constexpr int twiddle_entries();

// Cooperative copy of the twiddle table into LDS (visible after the caller's next barrier).
template <int S, int NT>
__device__ __forceinline__ void load_twiddles(float2 *__restrict__ tw_lds, const float2 *__restrict__ tw, int tid) {
    for (int i = tid; i < twiddle_entries<S>(); i += NT) tw_lds[lds_phys(i)] = tw[i];
}

// fp32 stage tables (same layout as the fp16 ones below, unscaled): linear copy into LDS
template <int S, int NT>
__device__ __forceinline__ void load_stage_tables(float2 *__restrict__ tw_lds, const float2 *__restrict__ tab, int tid);

// fp16 stage-twiddle tables: for every stage after the first, [q][k] (q < R, k < NS) entries
// stage_scale(R) * exp(+2 pi j q k / (NS R)) (inverse direction), concatenated in stage order.
template <int S>
__host__ __device__ constexpr int half_table_entries();

// Radix lists of the supported lengths (must equal factor_radices() on the host: 16s first).
template <int S>
struct TPlan;
#define BDS_TPLAN(S_, ...)                                                                         \
    template <>                                                                                    \
    struct TPlan<S_> {                                                                             \
        template <int T, int NT, int DIR, bool TAB = false, class C, class Src, class Dst>         \
        __device__ __forceinline__ static void run(C *buf, const C *tw, int tid, Src src, Dst dst) { \
            tfft_run<C, S_, T, NT, DIR, 1, 0, false, TAB, Src, Dst, NoHook, __VA_ARGS__>(buf, tw, tid, src, dst, NoHook{}); \
        }                                                                                          \
        /* first stage with adjacent lanes on adjacent transforms (tstage JFAST) */                \
        template <int T, int NT, int DIR, class C, class Src, class Dst, class Hook>               \
        __device__ __forceinline__ static void run_jfast(C *buf, const C *tw, int tid, Src src, Dst dst, Hook hook) { \
            tfft_run<C, S_, T, NT, DIR, 1, 0, true, false, Src, Dst, Hook, __VA_ARGS__>(buf, tw, tid, src, dst, hook); \
        }                                                                                          \
        template <int T, int NT, int DIR, bool TAB = false, class C, class Src, class Dst, class Hook> \
        __device__ __forceinline__ static void run_hook(C *buf, const C *tw, int tid, Src src, Dst dst, Hook hook) { \
            tfft_run<C, S_, T, NT, DIR, 1, 0, false, TAB, Src, Dst, Hook, __VA_ARGS__>(buf, tw, tid, src, dst, hook); \
        }                                                                                          \
        static constexpr int kRadix[] = {__VA_ARGS__};                                             \
    };
BDS_TPLAN(256, 16, 16)
BDS_TPLAN(512, 16, 16, 2)
BDS_TPLAN(768, 16, 16, 3)
BDS_TPLAN(1024, 16, 16, 4)
BDS_TPLAN(1280, 16, 16, 5)
BDS_TPLAN(2048, 16, 16, 8)
BDS_TPLAN(3072, 16, 16, 4, 3)
BDS_TPLAN(4096, 16, 16, 16)
#undef BDS_TPLAN

// A stage of radix R behind NS points reads W_S^(m k TWS), k < NS, TWS = S / (NS R), with m up to 8 (radix 16:
// the power-of-two multiples), 4 (radix 8) or R - 1 (the other radices): the table only needs the entries up to
// the largest such index (768 = 16 16 3: 511 of 768 entries; all-16/8 plans: under half).
template <int S>
__host__ __device__ constexpr int twiddle_entries() {
    int ns = 1, top = 0;
    for (int r : TPlan<S>::kRadix) {
        if (ns > 1) {
            const int m = r == 16 ? 8 : r == 8 ? 4 : r - 1;
            const int idx = m * (ns - 1) * (S / (ns * r));
            top = idx > top ? idx : top;
        }
        ns *= r;
    }
    return top + 1;
}

template <int S>
__host__ __device__ constexpr int half_table_entries() {
    int ns = 1, n = 0;
    for (int r : TPlan<S>::kRadix) {
        if (ns > 1) n += ns * r;
        ns *= r;
    }
    return n;
}

template <int S, int NT>
__device__ __forceinline__ void load_stage_tables(float2 *__restrict__ tw_lds, const float2 *__restrict__ tab, int tid) {
    for (int i = tid; i < half_table_entries<S>(); i += NT) tw_lds[i] = tab[i];
}

}  // namespace bds
