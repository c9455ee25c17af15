// Inverse column pass of the search with wave-private transforms (gfx950): the default column pass of the
// fp32-arithmetic search on the specialised plans (replaces k_cols_inv_max_f of bds_acq_f32.h there).
//
// What it computes is what that kernel computes -- the L1-point inverse transforms down the columns of the
// inter-pass buffer, w_d |y_d| + w_p |y_p| per lag, and the sieve's candidates -- but organised around the
// measured costs of this chip (tools/probe/valu_rate.hip): a workgroup barrier between every stage of an LDS
// transform (twelve per tile in the old kernel: 39 % of its wave-cycles were waits), LDS writes at a third of
// the read rate, 4-cycle conversions and 8-cycle square roots.
//
//   S = 64 R1 points per column (R1 = 4, 8, 12, 16), a tile of 8 adjacent columns per workgroup of 4 waves.
//   X[p + R1 (u + 8 v)] = sum_bl w8^(bl v) w64^(bl u) sum_bh w8^(bh u) [ w_S^(b p) sum_q w_R1^(q p) x[b + 64 q] ],  b = bl + 8 bh
//
//   phase A (cooperative, first stage fed straight from global memory): wave i, lane (cp = lane & 3, bq = lane >> 2)
//       takes butterfly b = 16 i + bq of column pair cp: 8 bytes per row, four lanes share a 32-byte piece of a
//       tile row; radix-R1 in registers, twiddle w_S^(b p) from per-lane constants, result to the LDS region
//       of the wave that owns column pair cp at [m = c R1 + p][b]                      -- barrier --
//   phase B (wave w owns columns 2w, 2w+1 and its LDS region; no workgroup barrier inside): lane (ml = lane & 7,
//       bl = lane >> 3), slots s: m = ml + 8 s.  Radix-8 over bh, twiddle w64^(bl u) from per-lane constants, written
//       back IN PLACE at [m][8 u + bl]; then lane (ml, u) reads row u, radix-8 over bl, magnitudes in registers.
//       (Columns are swapped inside aligned quads by (ml >> 1) -- struct WCols -- which keeps every access class conflict-free.)
//   Two LDS round trips per point instead of three-plus-tables, three barriers per tile and two components
//   instead of twelve-plus, every stage twiddle a per-lane constant (round 4: TWO coalesced loads from a per-lane table,
//   w_S^b and w_64^u -- the other sixteen are their powers: a load instruction costs a wave ~200 cycles of issue time here),
//   the rows of both components in flight before anything else happens.
//   Every LDS access class is bank-conflict free (tools/proto_cols_wave.py models the layouts lane by lane).
//
// Outputs (no per-tile records): per cell the packed maximum {value, first lag} by a 64-bit atomic max (rare:
// only when a wave beats the cell's current value), per PRN a running lower bound `lb` of the sieve maximum,
// and the candidate list: every lag whose value is within `keep` of max(wave maximum, lb).  Both are lower
// bounds of the PRN's final maximum M, so every lag >= keep * M is on the list -- the completeness argument of
// DESIGN.md section 1.5 -- while the list stays a few thousand entries per PRN instead of one record per tile.
#pragma once

#include "bds_acq_f32.h"
#include "bds_fft_pk.h"
#include "bds_lds.h"

// Timing experiments (tools/exp/exp_wparts.sh; results are INVALID with any of these defined):
//   BDS_EXP_WC_NOTAIL  nothing after the wave maximum (no bounds, no list, no atomics)
//   BDS_EXP_WC_NOLOAD  the inter-pass buffer is not read
//   BDS_EXP_WC_NOBAR   workgroup barriers compiled out
#ifdef BDS_EXP_WC_NOBAR
#define BDS_WSYNC() __builtin_amdgcn_s_waitcnt(0)
#else
#define BDS_WSYNC() __syncthreads()
#endif

// cache policy of the column pass's tile-row loads (aux of the buffer load: 0 default, 2 = nt; round 5, tools/exp/r5_nt.sh)
#ifndef BDS_COLS_AUX
#define BDS_COLS_AUX 0
#endif

namespace bds {

template <int DIR>
struct Butterfly<12, DIR> {
    // n = 3 a + b, k = c + 4 d:  X[c + 4 d] = sum_b W3^(b d) W12^(b c) sum_a x[3 a + b] W4^(a c)
    __device__ __forceinline__ static void run(float2 *v) {
        const float h = 0.5f, s = 0.86602540378443864676f;
        float2 t[3][4];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[b][0] = v[b];
            t[b][1] = v[b + 3];
            t[b][2] = v[b + 6];
            t[b][3] = v[b + 9];
            Butterfly<4, DIR>::run(t[b]);
        }
        t[1][1] = mulc<DIR>(t[1][1], s, h);    // W12^1
        t[1][2] = mulc<DIR>(t[1][2], h, s);    // W12^2
        t[1][3] = rot90<DIR>(t[1][3]);         // W12^3 = -j
        t[2][1] = mulc<DIR>(t[2][1], h, s);    // W12^2
        t[2][2] = mulc<DIR>(t[2][2], -h, s);   // W12^4
        t[2][3] = make_float2(-t[2][3].x, -t[2][3].y);  // W12^6 = -1
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float2 u[3] = {t[0][c], t[1][c], t[2][c]};
            Butterfly<3, DIR>::run(u);
            v[c] = u[0];
            v[c + 4] = u[1];
            v[c + 8] = u[2];
        }
    }
};

// geometry of the wave-private column pass for a length-S transform
template <int S>
struct WCols {
    static_assert(S % 64 == 0 && (S / 64 == 4 || S / 64 == 8 || S / 64 == 12 || S / 64 == 16), "S = 64 x {4, 8, 12, 16}");
    static constexpr int R1 = S / 64;     // radix of the first stage
    static constexpr int M = 2 * R1;      // (column of the pair, p) combinations a wave owns
    static constexpr int SL = M / 8;      // radix-8 butterflies per lane and stage
    // LDS layout of a wave's region: element (m, column) at [m MS + (column ^ ((m & 7) >> 1))]; regions RS apart.
    // The hardware serves a ds_read_b64 in halves of 32 lanes over 64 banks (32 eight-byte slots) and a ds_write_b64 in groups
    // of 16 contiguous lanes over 32 banks (16 slots).  The phase-B readers (8 ml x 4 bl or u per half) need MS = 4 (mod 32),
    // the in-place writers (8 ml x 2 bl per group) MS = 2 (mod 4): the plain layout of rounds 3-4 read clean and wrote 2-way
    // (SQ_LDS_BANK_CONFLICT was 46 % of this kernel's LDS cycles).  Swapping columns inside every aligned quad by (ml >> 1)
    // separates the writers ml and ml + 4, keeps the stage-2 reads a permutation of what they were, and makes the stage-3
    // reads clean WITHOUT the rotated start they used to need; phase A's writers (4 column pairs x 4 b per group) want the
    // regions 4 (mod 16) slots apart.  tools/proto_cols_wave.py checks every access class under this bank model.
    static constexpr int MS = 68;
    static constexpr int RS = M * MS + 4;
    static constexpr int NW = 4, NT = 256, T = 8;
    static constexpr size_t kLdsBytes = sizeof(float2) * NW * RS;
    // waves per SIMD the register budget is set for (measured unconstrained need: 80 / 123 / 174 / 215 VGPRs; the
    // LDS regions allow 9 / 4 / 3 / 2 workgroups per CU, so neither resource is wasted on the other's account)
#ifdef BDS_WCOLS_OCC
    static constexpr int kOcc = BDS_WCOLS_OCC;
#else
    static constexpr int kOcc = R1 == 4 ? 5 : R1 == 8 ? 4 : R1 == 12 ? 3 : 2;
#endif
};

struct WColsArgs {
    const float2 *wtab;  // per-lane twiddle table of the plan (wcols_table_entries<S>())
    int L2, ntiles, G, n_items;
    const void *Bw;
    long L;
    float w0, w1;
    int lo1, hi1, lo2, hi2;
    const int4 *cell_rng;            // optional per-cell (lo1, hi1, lo2, hi2), MASKED kernels only
    unsigned long long *cellmax;     // [run-wide cell]: (value bits << 32) | ~lag, by atomic max
    float *lb;                       // [(run-wide cell) / lb_div]: running lower bound of that PRN's sieve maximum
    int lb_div;
    Extra *extra;                    // candidate list
    int *extra_count;
    int extra_cap;
    int cell0;                       // run-wide index of cell 0 of this launch
    float keep;                      // 1 - tolerance of the sieve
    int qchunk;                      // adjacent quads (4 tiles = one 128-byte line per row) of a cell that follow each other in the list
    unsigned long long *clk;         // optional (BDS_ACQ_CLOCKPROBE): [2], [3] += shader-clock / reference-clock ticks of sampled workgroups
};

__device__ __forceinline__ unsigned long long wc_pack(float v, int lag) {
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(~(unsigned)lag);
}

// entries of the per-lane twiddle table of a length-S plan: (R1 - 1) x 256 for phase A (w_S^(b p), p = 1 .. R1 - 1, indexed
// [p - 1][thread]) followed by 7 x 64 for phase B (input j of lane (ml, u = lane / 8) in stage 3 is bl = (j + u) & 7:
// w_64^(u (bl - u)), j = 1 .. 7, indexed [j - 1][lane]); inverse direction
template <int S>
__host__ __device__ constexpr int wcols_table_entries() {
    return (WCols<S>::R1 - 1) * WCols<S>::NT + 7 * 64;
}

// ILV: the inter-pass buffer holds both components of an element side by side ([cell][element][component], written so by
// k_rows_wave_f<2, true>): one 16-byte load per lane and tile row fetches both (64-byte pieces per tile row instead of 32-byte
// ones, half the load instructions)
// PK: butterflies and twiddle products on packed fp32 pairs (bds_fft_pk.h)
template <int S, int NCOMP, bool MASKED, class ST, int NV, bool ILV = false, bool PK = false>
__global__ __launch_bounds__(WCols<S>::NT, WCols<S>::kOcc) void k_cols_wave_f(WColsArgs A) {
    using C = typename std::conditional<PK, v2f, float2>::type;
    static_assert(NV >= 1 && NV <= 8, "outputs of the last radix-8 stage");
    static_assert(!ILV || (NCOMP == 2 && std::is_same<ST, __half2>::value), "interleaved components: two, fp16 storage");
    using W = WCols<S>;
    constexpr bool HS = std::is_same<ST, __half2>::value;
    constexpr int R1 = W::R1, SL = W::SL, MS = W::MS, RS = W::RS;
    extern __shared__ __attribute__((aligned(16))) float2 ldsf[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L2 = A.L2;
    const long L = A.L;
    const ClockProbe clkp(A.clk ? A.clk + 2 : nullptr, 255);
    PH_DECL(24);

    // ---- the item of this workgroup ----------------------------------------------------------------------
    // One tile per workgroup, workgroups started by the hardware in list order.  Workgroup id % 8 = XCD; XCD x keeps the
    // contiguous run of tiles [x TX, (x + 1) TX) of every cell.  Its list: four adjacent tiles of one cell (they share
    // 128-byte lines and are loaded within microseconds of each other by four workgroups that start together) and the next
    // qchunk - 1 such quads (512 contiguous bytes per row with the default 4: DRAM page locality -- 1.70 ms per cfg3 launch
    // against 1.84 with single lines), then the same of the NEXT cell (not the cell's next tiles), each XCD starting one
    // eighth of the way further round the cells: the workgroups that run at the same time then work on different cells
    // (~16 per cell), so that a cell's running maximum is settled by a few early waves instead of every wave of the cell
    // seeing it unset at once.
    // Measured alternatives (cfg3, per 201-cell launch): a persistent grid with a static item -> workgroup map drifts apart
    // over its ~130 items and every tile refetches its lines (2.7x the HBM traffic, 2.31 ms against 1.80); persistent
    // workgroups drawing tickets from a per-XCD atomic counter stay in order, but same-line device-scope atomics complete
    // at ~10 M/s (9.6 ms); 2 - 8 items per workgroup with the next tile's rows prefetched: 2.16 - 2.29 ms.
    const int TX = A.ntiles >> 3;
    int g, c0;
    {
        const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        const int t4 = j & 3, rest = j >> 2;
        const int qi = rest % A.qchunk, rest2 = rest / A.qchunk;  // qchunk adjacent quads of a cell before the next cell
        const int tgc = rest2 / A.G, gi = rest2 - tgc * A.G;
        const int tg = tgc * A.qchunk + qi;
        g = gi + xcd * (A.G >> 3);
        g = g >= A.G ? g - A.G : g;
        c0 = (xcd * TX + tg * 4 + t4) * W::T;
    }
#ifdef BDS_EXP_PHASES
    asm volatile("" : "+s"(g), "+s"(c0));
#endif
    PH_MARK(16);  // kernel arguments read, item decoded

    // ---- rows of both components: in flight before anything else ---------------------------------------------
    // phase A: butterfly b of column pair cp takes rows b + 64 q
    const int cp = lane & 3, b = 16 * wave + (lane >> 2);
    using Raw = typename std::conditional<ILV, uint4, typename std::conditional<HS, uint2, float4>::type>::type;
    // (buffer loads: the descriptor of (cell, component, tile) and the row offsets live in scalar registers, the lane
    //  offset is one VGPR -- global_load with per-row 64-bit vector addresses cost two dozen VGPRs and their arithmetic;
    //  a flat_load would also count on the LDS counter and every LDS wait of the transform would wait for the rows)
    const int voff = (b * L2 + 2 * cp) * (int)sizeof(ST) * (ILV ? 2 : 1);  // < 2^23
    auto fetch = [&](Raw(&pre)[R1], int comp) {
        const char *base = ILV ? (const char *)A.Bw + ((long)g * L + c0) * (long)(2 * sizeof(ST))
                               : (const char *)A.Bw + (((long)g * NCOMP + comp) * L + c0) * (long)sizeof(ST);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
        const int rowstep = 64 * L2 * (int)sizeof(ST) * (ILV ? 2 : 1);
#pragma unroll
        for (int q = 0; q < R1; ++q) {
#ifdef BDS_EXP_WC_NOLOAD
            if (A.w0 != 1.2345f) {
                pre[q] = Raw{};
                continue;
            }
#endif
            if constexpr (ILV) {
#ifdef BDS_COLS_DMA
                // experiment: the tile rows by LDS DMA -- buffer_load_dwordx4 ... lds, lane i's 16 bytes at
                // [q][i] of the wave's own region (free until phase A writes: one tile per workgroup) -- read back below
                typedef __attribute__((address_space(3))) void *lds_vp;
                const unsigned off = __builtin_amdgcn_readfirstlane(lds_offset(reinterpret_cast<C *>(ldsf) + wave * RS) + q * 1024u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_vp)(uintptr_t)off, 16, voff, q * rowstep, 0, BDS_COLS_AUX);
                pre[q] = make_uint4(0, 0, 0, 0);
#else
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, q * rowstep, BDS_COLS_AUX);
                pre[q] = make_uint4(v[0], v[1], v[2], v[3]);  // (column 2 cp: data, pilot; column 2 cp + 1: data, pilot)
#endif
            } else if constexpr (HS) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, q * rowstep, BDS_COLS_AUX);
                pre[q] = make_uint2(v[0], v[1]);
            } else {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, q * rowstep, 0);
                pre[q] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
        }
    };
    // Order of the requests (round 4): a load instruction costs this wave 120 - 290 cycles of issue time while the other
    // workgroups of the CU keep the vector-memory unit busy (tools/phases.py: 8.3 k of a wave's 24 k cycles went into requesting
    // 45 loads before anything was computed).  So only what phase A of component 0 needs is requested up front -- the rows
    // and ONE twiddle w_S^b, the others being its powers (ten complex products; their rounding, a few 1e-7, is far inside
    // the sieve's tolerance) -- and the rest (component 1's rows unless interleaved, the phase-B constants, the running bounds)
    // goes out behind phase A's arithmetic, where its latency is covered by phase B of component 0.
    Raw pre0[R1], pre1[(NCOMP > 1 && !ILV) ? R1 : 1];
    C twA[R1];  // w_S^(b p), inverse direction
    C twB1;  // w_64^u, the first of the phase-B constants (the others are its powers, formed behind phase A)
    {
        const float2 t = A.wtab[tid];
        cx_set(twA[1], t.x, t.y);
        const float2 u = A.wtab[(R1 - 1) * W::NT + lane];
        cx_set(twB1, u.x, u.y);
    }
    fetch(pre0, 0);
    PH_MARK(17);  // rows of component 0 requested
    const int cell = A.cell0 + g;
    float *const lbp = A.lb + cell / A.lb_div;
    {
        const C w = twA[1];
#pragma unroll
        for (int p = 2; p < R1; ++p) twA[p] = (p & 1) ? cx_mul(twA[p - 1], w) : cx_mul(twA[p / 2], twA[p / 2]);
    }
    PH_MARK(19);  // phase-A constants formed
    C *const ldsc = reinterpret_cast<C *>(ldsf);
    C *wrA[4];  // + m MS, by the swizzle (m & 7) >> 1 of row m
#pragma unroll
    for (int k = 0; k < 4; ++k) wrA[k] = ldsc + cp * RS + (b ^ k);
    // phase B: stage 2 as lane (ml, bl), stage 3 as lane (ml, u) with u = bl
    const int ml = lane & 7, bl = lane >> 3;
    // stage-2 twiddle w_64^(bl u), applied by stage 3 to its INPUTS (bds_fft_fma.h: folded into the first butterfly layer): input
    // j of lane (ml, u) is bl = j
    C twB[8];
    C *const rw2 = ldsc + wave * RS + ml * MS + (bl ^ (ml >> 1));  // + s 8 MS + 8 bh (read), + 8 u (write back)
    const unsigned rw2a = lds_offset(rw2);
    unsigned rd3a[8];                                              // row u: + s 8 MS
#pragma unroll
    for (int j = 0; j < 8; ++j) rd3a[j] = lds_offset(ldsc + wave * RS + ml * MS + 8 * bl + (j ^ (ml >> 1)));
    // lag of output (s, v) of this lane: row e = p + R1 u + 8 R1 v of column 2 wave + c (m = ml + 8 s = c R1 + p), i.e.
    // lbase[s] + v vstep                                                                            (L < 2^31)
    int lbase[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int m = ml + 8 * s, c = m >= R1 ? 1 : 0;
        lbase[s] = (m - c * R1 + R1 * bl) * L2 + c + 2 * wave + c0;
    }
    const int vstep = 8 * R1 * L2;
    int lo1 = A.lo1, hi1 = A.hi1, lo2 = A.lo2, hi2 = A.hi2;
    if (MASKED && A.cell_rng) {
        const int4 r = A.cell_rng[g];
        lo1 = r.x, hi1 = r.y, lo2 = r.z, hi2 = r.w;
    }

    // first stage of both columns of the pair, twiddled
    auto phaseA = [&](const Raw(&pre)[R1], C(&z)[2][R1], int comp) {
        auto set = [](C &d, float2 t) { cx_set(d, t.x, t.y); };
#pragma unroll
        for (int q = 0; q < R1; ++q) {
            if constexpr (ILV) {
                set(z[0][q], h2_to_f2(comp == 0 ? pre[q].x : pre[q].y));
                set(z[1][q], h2_to_f2(comp == 0 ? pre[q].z : pre[q].w));
            } else if constexpr (HS) {
                set(z[0][q], h2_to_f2(pre[q].x));
                set(z[1][q], h2_to_f2(pre[q].y));
            } else {
                cx_set(z[0][q], pre[q].x, pre[q].y);
                cx_set(z[1][q], pre[q].z, pre[q].w);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            cx_bfly<R1>(z[c]);
#pragma unroll
            for (int p = 1; p < R1; ++p) z[c][p] = cx_mul(z[c][p], twA[p]);
        }
    };
    // opaque to the scheduler: the raw rows are "produced" where this stands, nothing consuming them moves above it
    auto pin = [](Raw(&pre)[R1]) {
#pragma unroll
        for (int q = 0; q < R1; ++q) {
            if constexpr (HS && !ILV)
                asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y));
            else
                asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y), "+v"(pre[q].z), "+v"(pre[q].w));
        }
    };
    auto wave_sync = [] {  // LDS traffic of one wave is in order; this only stops the compiler from moving it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    float sq[NCOMP][SL][NV];  // |y|^2 per component
    float bmax = 0.f;         // maximum of |y_d|^2 (+ |y_p|^2) over the wave's outputs
    float lbv = 0.f;
    unsigned cur = 0;
    PH_MARK(15);  // set-up: kernel arguments, item, loads issued
    PH_WAIT_VM();
#ifdef BDS_COLS_DMA
    if constexpr (ILV) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 *stage = reinterpret_cast<const uint4 *>(reinterpret_cast<C *>(ldsf) + wave * RS) + lane;
#pragma unroll
        for (int q = 0; q < R1; ++q) pre0[q] = stage[64 * q];
        __syncthreads();  // every wave has its rows in registers before phase A writes into the regions
    }
#endif
    PH_MARK(0);  // rows of both components + per-lane constants have arrived
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
        {
            C z[2][R1];
            if (comp == 0) {
                phaseA(pre0, z, 0);
                // the deferred requests (see above): answered while phase B of this component runs
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NCOMP > 1 && !ILV) fetch(pre1, 1);
                // w_64^(u j) as powers of w_64^u: six complex products instead of six more load instructions, each of which
                // costs this wave ~200 cycles of issue time here (round 4, late: the phase clocks showed component 0's phase A
                // at 2.7 k cycles against 1.1 k for component 1's)
                twB[1] = twB1;
#pragma unroll
                for (int j = 2; j < 8; ++j) twB[j] = (j & 1) ? cx_mul(twB[j - 1], twB1) : cx_mul(twB[j / 2], twB[j / 2]);
                // the cell's maximum so far and the PRN's running bound (L2 / fabric latency), used after the transforms;
                // stale values are lower values, which only costs a redundant visit of the rare path below
                lbv = __hip_atomic_load(lbp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cur = (unsigned)(__hip_atomic_load(A.cellmax + cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // (pinned here: moved up into the first component's last stage it doubles the live registers)
                if constexpr (ILV) {
                    pin(pre0);
                    phaseA(pre0, z, 1);
                } else if constexpr (NCOMP > 1) {
                    pin(pre1);
                    phaseA(pre1, z, 1);
                }
            }
            PH_MARK(1 + 6 * comp);  // phase A arithmetic
            if (comp > 0) BDS_WSYNC();  // every wave is through with its region (last reads of the previous component)
            PH_MARK(2 + 6 * comp);  // (barrier before the writes)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int p = 0; p < R1; ++p) wrA[((c * R1 + p) & 7) >> 1][(c * R1 + p) * MS] = z[c][p];
            }
            PH_WAIT_LGKM();
            PH_MARK(3 + 6 * comp);  // phase A writes issued and landed
        }
        BDS_WSYNC();
        PH_MARK(4 + 6 * comp);  // barrier
        // ---- phase B, one slot (= 8 of the wave's rows m) at a time so that only 16-32 points are live:
        //   st2(s): radix 8 over bh, back in place;  st3(s): twiddle, radix 8 over bl, magnitudes.
        // Row m is read and written by the 8 lanes of one ml only, all in this wave, and LDS traffic of a wave is in
        // order: st2(s) may write as soon as its own reads are in, st3(s) may read as soon as st2(s) has written.
        // The units are software-pipelined by hand (st2(s + 1) sits between the write and the read-back of slot s)
        // and fenced, so that the scheduler neither serialises the LDS latency nor hoists every read to the top.
        // (reads: bds_lds.h -- single ds_read_b64 instructions, a batch and its wait per statement)
        auto st2 = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            C y[8];
            lds_read8<s * 8 * MS * (int)sizeof(C), 8 * (int)sizeof(C)>(y, rw2a);
            wave_sync();
            cx_bfly8<false>(y, (const C *)nullptr);
#pragma unroll
            for (int u = 0; u < 8; ++u) rw2[s * 8 * MS + 8 * u] = y[u];
            wave_sync();
        };
        auto st3 = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            C y[8];
            lds_read8p<s * 8 * MS * (int)sizeof(C)>(y, rd3a);
            wave_sync();
            cx_bfly8<true>(y, twB);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                // Only |y|^2 is formed here; the square roots, the lag masks and the exact maximum belong to the (rare)
                // tail below.  (And no branches in here, not even uniform ones: every basic-block boundary pins the
                // butterfly's outputs in registers -- 254 VGPRs with a workgroup-uniform "row range searched at all?"
                // test per v against 174 without.  Output v covers rows 8 R1 v .. 8 R1 (v + 1) - 1; the host instantiates
                // NV = 6 when no searched lag lies beyond row 48 R1 -- the padded transform is ~1.6 N long -- and the
                // unused outputs of the last butterfly fall away at compile time.)
                const float2 t = cx_f2(y[v]);
                const float q2 = t.x * t.x + t.y * t.y;
                sq[comp][s][v] = q2;
                if (comp == NCOMP - 1) bmax = fmaxf(bmax, NCOMP > 1 ? sq[0][s][v] + q2 : q2);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        st2(std::integral_constant<int, 0>{});
        static_for<SL>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s + 1 < SL) st2(std::integral_constant<int, s + 1>{});
            st3(sc);
        });
        PH_MARK(5 + 6 * comp);  // phase B
    }
    // ---- maximum of the wave's two columns, candidates ------------------------------------------
    // Cauchy-Schwarz: (w_d |y_d| + w_p |y_p|)^2 <= (w_d^2 + w_p^2) (|y_d|^2 + |y_p|^2).  If even that bound, over all of the
    // wave's outputs, stays below both the cell's maximum so far and the sieve threshold of the PRN's running bound, the
    // wave has nothing to report: no square root was taken and no lag was formed -- the common case once a cell's first
    // few workgroups are through.  (1e-5: rounding of the bound and of the squares, 40 x the fp32 unit.)
    const float wsum2 = NCOMP > 1 ? A.w0 * A.w0 + A.w1 * A.w1 : A.w0 * A.w0;
    const float bw = wave_max_f32(bmax) * wsum2 * 1.00001f;
    const float curv = __uint_as_float(cur), lim = fminf(curv, lbv * A.keep);
#ifdef BDS_EXP_WC_NOTAIL
    if (bw == 1.2345f) {
#else
    if (!(bw < lim * lim)) {  // (wave-uniform; also taken while the bounds are unset or not finite)
#endif
        // exact values; lags outside the searched ranges hold -1 (searched values are >= 0)
        float mag[SL][NV];
        float mx = -1.f;
#pragma unroll
        for (int s = 0; s < SL; ++s) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                // raw v_sqrt_f32 (1 ulp): the value only feeds the sieve
                float a = A.w0 * __builtin_amdgcn_sqrtf(sq[0][s][v]);
                if constexpr (NCOMP > 1) a += A.w1 * __builtin_amdgcn_sqrtf(sq[NCOMP - 1][s][v]);
                const int lag = lbase[s] + v * vstep;
                const bool ok = MASKED ? ((lag >= lo1 && lag <= hi1) || (lag >= lo2 && lag <= hi2)) : lag <= hi1;
                mag[s][v] = ok ? a : -1.f;
                mx = fmaxf(mx, mag[s][v]);
            }
        }
        const float Mw = wave_max_f32(mx);
        if (Mw >= 0.f) {  // (wave-uniform) something of these two columns is searched
        const float thr = fmaxf(Mw, lbv) * A.keep;
        const bool newmax = __float_as_uint(Mw) >= cur;  // this wave holds (a tie of) the cell's maximum so far
        if (newmax || __builtin_amdgcn_ballot_w64(mx >= thr) != 0) {
            // Rare (wave-uniform): the values go through the wave's own LDS region (nobody else touches it any more)
            // and a compact loop picks the maximum's first lag and every lag within the sieve tolerance of the bound.
            float *sm = reinterpret_cast<float *>(ldsf + wave * RS) + lane;  // [k = 8 s + v][lane]
            wave_sync();
#pragma unroll
            for (int s = 0; s < SL; ++s) {
#pragma unroll
                for (int v = 0; v < 8; ++v) sm[(8 * s + v) * 64] = v < NV ? mag[s][v < NV ? v : 0] : -1.f;
            }
            wave_sync();
            auto lag_at = [&](int k) {
                const int m = ml + (k & ~7), c = m >= R1 ? 1 : 0;
                return (m - c * R1 + R1 * bl + 8 * R1 * (k & 7)) * L2 + c0 + 2 * wave + c;
            };
            int best = 0x7fffffff, total = 0;
#pragma nounroll
            for (int k = 0; k < 8 * SL; ++k) {
                const float a = sm[k * 64];
                total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(a >= thr));
                if (newmax && a == Mw) best = min(best, lag_at(k));
            }
            if (total > 0) {  // one reservation per wave on the list's counter
                int base = 0;
                if (lane == 0) base = atomicAdd(A.extra_count, total);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma nounroll
                for (int k = 0; k < 8 * SL; ++k) {
                    const float a = sm[k * 64];
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(a >= thr);
                    if (a >= thr) {
                        const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        BDS_DASSERT(idx >= 0 && lag_at(k) >= 0 && (long)lag_at(k) < L && cell >= A.cell0 && cell < A.cell0 + A.G);
                        if (idx < A.extra_cap) {
                            Extra ex;
                            ex.v = a;
                            ex.lag = lag_at(k);
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
    PH_MARK(13);  // tail
    PH_FLUSH(0, 24);
    clkp.finish(tid);
}

}  // namespace bds
