// Inverse row pass of the search for 4096-point rows with wave-private stages (gfx950); an alternative to k_rows_inv_f of
// bds_acq_f32.h on the x 4096 plans (cfg3: 768 x 4096), selected by BDS_ACQ_WROWS.
//
// Same job and same HBM layout as that kernel -- spectrum product X_b .* conj(C_p), inverse rows (length 4096), inter-pass
// twiddle, fp16 store (fp16 storage only); one workgroup owns spectrum row k1 for up to GC cells, code-spectrum rows and twiddles set up
// once, the next cell's spectrum row in flight -- organised like the wave-private column pass (bds_acq_wcols.h), decimated in
// frequency so that the LAST stage leaves thread e'' the outputs e'' + 256 p' (coalesced stores; the 16-byte pieces are on the
// loads, which the L1 / L2 completes to full lines):
//
//   r = 16 b' + q',  b' = bl + 16 bh;   e = e'' + 256 p',  e'' = u + 16 v
//   Y_q'[u + 16 v] = sum_bl w16^(bl v) w256^(bl u) sum_bh w16^(bh u) x[16 (bl + 16 bh) + q']      wave w: q' = 4w .. 4w+3
//   X[e'' + 256 p'] = sum_q' w16^(q' p') w4096^(q' e'') Y_q'[e'']                                  thread e''
//
//   phase 1a: lane (ql = lane & 3, bl = lane >> 2) forms its 16 products x[16 bl + q' + 256 bh] from registers, radix 16 over
//       bh, to the wave's own LDS region at [64 u + lane]
//   phase 1b: lane (ql, u = lane >> 2) reads row u starting at column u (bank-conflict free; a rotation of the butterfly's
//       inputs = the factor w16^(-u v) on its outputs, folded into the per-lane inter-pass twiddle), twiddle w256^(bl u) from
//       per-lane constants ON THE INPUTS (bds_fft_fma.h: folded into the first butterfly layer), radix 16 over bl, Y to the
//       exchange buffer at [20 e'' + ((e'' >> 3) & 3) + q']                                                            -- barrier --
//   phase 2 : thread e'' reads its 16 q', twiddle w4096^(q' e'') from per-lane constants, radix 16, inter-pass twiddle, store.
//   Against k_rows_inv_f: every stage twiddle is a per-lane constant (30 of them were rebuilt from four table reads with
//   eleven complex products per butterfly and twiddled stage: 176 of its 1693 vector instructions per cell), no LDS twiddle
//   table, 4 workgroup barriers per cell instead of 7, no LDS bank conflicts.
//   tools/proto_rows_wave.py models the stage algebra, the lane maps and the LDS layout (conflict-free in every access class).
#pragma once

#include "bds_acq_f32.h"
#include "bds_fft_pk.h"
#include "bds_lds.h"

namespace bds {

// per-lane twiddle table of the 4096-point row pass: 15 x 256 for phase 2 (w4096^(q' e''), q' = 1 .. 15, [q' - 1][thread])
// followed by 15 x 64 for phase 1b (input j of lane (ql, u = lane >> 2) is bl = (j + u) & 15: w256^(u (bl - u)), j = 1 .. 15,
// [j - 1][lane]), the 16 values w16^k and the 16 values w256^(u u) (the factor taken out of the phase-1b twiddles so that
// input 0 needs none; it goes into the inter-pass twiddle); inverse direction
constexpr int kWRowsTableEntries = 15 * 256 + 15 * 64 + 16 + 16;
constexpr int kWRowsRegion = 1024;  // elements of a wave's region (4 q' x 256)
// Exchange buffer: Y_q'[e''] at [20 e'' + ((e'' >> 3) & 3) + q'].  The hardware serves a ds_write_b64 in groups of 16 contiguous
// lanes over 32 banks (16 eight-byte slots) and a ds_read_b64 in halves of 32 lanes over 64 banks (32 slots): the writers of a
// group -- 4 q' x 4 consecutive u -- need the stride = 4 (mod 16 slots), the readers -- 32 consecutive e'' -- an odd one, so no
// plain stride serves both (19, rounds 3-4, read clean and wrote 2-way: SQ_LDS_BANK_CONFLICT was 22 % of the kernel's LDS
// cycles).  Stride 20 with every octet of e'' shifted by one slot more (mod 4) is clean both ways, and both sides still
// address it as one per-lane base + a compile-time offset.
constexpr int kWRowsXS = 20;
// cache policy of the row pass's streams (round 5, tools/exp/r5_nt.sh, profiles/r05_nt_ab.txt; per 201 cells, alternating on one
// box): non-temporal STORES of the inter-pass buffer pair 3.175 vs 3.189 ms (kept: rows 1.685 vs 1.703, columns 1.451 vs 1.470);
// non-temporal loads of the signal-spectrum rows 3.40 vs 3.19 (rows 1.93 vs 1.70: dropped), of the code-spectrum rows as well
// 3.43; the column pass's tile-row loads with aux = 2 (nt) 3.34 (columns 1.63 vs 1.47: the 128-byte lines adjacent tiles share
// are evicted between them)
#ifndef BDS_ROWS_NT_X
#define BDS_ROWS_NT_X 0
#endif
#ifndef BDS_ROWS_NT_C
#define BDS_ROWS_NT_C 0
#endif
#ifndef BDS_ROWS_NT_ST
#define BDS_ROWS_NT_ST 1
#endif
constexpr bool kRowsNtX = BDS_ROWS_NT_X != 0, kRowsNtC = BDS_ROWS_NT_C != 0, kRowsNtSt = BDS_ROWS_NT_ST != 0;
constexpr size_t kWRowsLdsBytes = sizeof(float2) * (4 * kWRowsRegion + 256 * kWRowsXS + 4);

// PK: the butterflies and twiddle products on packed fp32 pairs (bds_fft_pk.h: half the vector issue slots for the same pipe time)
template <int NCOMP, bool ILV, bool PK>
__global__ __launch_bounds__(256, 2) void k_rows_wave_f(RowsFArgs A) {
    using C = typename std::conditional<PK, v2f, float2>::type;
    using ST = __half2;  // fp16 storage only: with fp32 storage the 32-byte load pieces cost more than the stages save (5.2 vs 3.1 ms)
    constexpr int S = 4096, XS = kWRowsXS;
    // both components of an element side by side in the inter-pass buffer ([cell][element][component]): the column pass fetches
    // a tile row of both with ONE 16-byte load per lane, this pass stores 8 bytes per lane and output -- half the vector-memory
    // instructions on either side (ILV; two components only)
    static_assert(!ILV || NCOMP == 2, "interleaved components: two of them");
    constexpr bool ILV_OK = ILV;
    extern __shared__ __attribute__((aligned(16))) float2 ldsf[];
    __shared__ float2 s_b[16];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long L = A.L;
    const ClockProbe clkp(A.clk, 63);
    PH_DECL(17);
    auto wave_sync = [] {  // LDS traffic of one wave is in order; this only stops the compiler from moving it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // per-lane stage twiddles of phase 1b (plan constants); those of phase 2 are completed per row below
    C twB[16];
#pragma unroll
    for (int u = 1; u < 16; ++u) {
        const float2 t = A.tw[15 * 256 + (u - 1) * 64 + lane];
        cx_set(twB[u], t.x, t.y);
    }
    const int ql = lane & 3, bl = lane >> 2;  // phase 1a: (ql, bl); phase 1b: (ql, u = bl)
    const int qp = 4 * wave + ql;             // this lane's q' in phase 1
    // its inputs: elements 16 bl + qp + 256 bh of the spectrum rows
    C *const ldsc = reinterpret_cast<C *>(ldsf);
    C *const wr1 = ldsc + wave * kWRowsRegion + lane;                     // + 64 u
    const C *const rd1 = ldsc + wave * kWRowsRegion + 64 * bl + ql;       // + 4 ((j + u) & 15)
    C *const wrx = ldsc + 4 * kWRowsRegion + XS * bl + (bl >> 3) + qp;    // e'' = u + 16 v: + 16 XS v + 2 (v & 1)
    const unsigned rd2a = lds_offset(ldsc + 4 * kWRowsRegion + XS * tid + ((tid >> 3) & 3));  // + q'

    for (int vb = (int)blockIdx.x; vb < A.nvb; vb += (int)gridDim.x) {
        const int xcd = vb & 7, m = vb >> 3;
        const int GC = A.GC, NCH = A.NCH;
        const int g0 = (m % NCH) * GC, k1 = (m / NCH) * 8 + xcd;
        const int g1 = g0 + GC < A.G ? g0 + GC : A.G;
        const ST *Cs = (const ST *)A.Cs;
        if (A.cell_cs) Cs += A.cell_cs[g0];
        if (tid < 16) s_b[tid] = A.twl.get<+1>((uint32_t)((long)k1 * 256 * tid));
        // spectrum row of the next cell, raw as stored: elements xoff + 256 bh
        // (stored in the order this kernel multiplies them, wrows_perm() of bds_acq_fast.h: element xoff + 256 bh of the row
        //  sits at [16 tid + bh] -- four 16-byte loads per lane, 4 KB contiguous per wave)
        uint32_t xn[16];
        auto fetch16 = [&](const ST *row, uint32_t(&d)[16], auto nt_c) {
            constexpr bool NT = decltype(nt_c)::value;
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            const u4v *p = reinterpret_cast<const u4v *>(row + 16 * tid);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // NT: read-once streams (the signal spectra: 2.5 GB per Doppler grid, nothing of it survives in L2 / MALL until
                // the next PRN's launch) are requested non-temporal -- BDS_ROWS_NT_X, measured in tools/exp/r5_nt.sh
                const u4v v = NT ? __builtin_nontemporal_load(p + k) : p[k];
                d[4 * k] = v.x, d[4 * k + 1] = v.y, d[4 * k + 2] = v.z, d[4 * k + 3] = v.w;
            }
        };
        auto fetch_x = [&](int g) {
            const int bin = A.cell_bin ? A.cell_bin[g] : A.bin0 + g;
            fetch16((const ST *)A.Xs + (long)bin * L + (long)k1 * S, xn, std::integral_constant<bool, kRowsNtX>{});
        };
        // The inter-pass twiddle W_L^(k1 e) of output e = tid + 256 p' factors into a per-thread part
        //   wi = W_L^(k1 tid) x storage scale x w16^(u v) (u = tid & 15, v = tid >> 4: undoes the rotated read of phase 1b)
        //        x w256^(u u) (the factor the phase-1b twiddles leave out)
        // and a workgroup-uniform part s_b[p'] = W_L^(256 k1 p').  The transform is linear: wi goes into the phase-2 INPUT
        // twiddles w4096^(q' tid) (once per row), s_b[] into scalar registers -- the 32 VGPRs a per-thread table of all 16
        // products took (round 3) hold the first component's packed outputs instead (interleaved stores).
        C twC[16];
        {
            float2 wi = A.twl.get<+1>((uint32_t)k1 * (uint32_t)tid);
            wi.x *= A.out_scale;
            wi.y *= A.out_scale;
            wi = cmul(wi, A.tw[15 * 256 + 15 * 64 + (((tid & 15) * (tid >> 4)) & 15)]);
            wi = cmul(wi, A.tw[15 * 256 + 15 * 64 + 16 + (tid & 15)]);
            cx_set(twC[0], wi.x, wi.y);
#pragma unroll
            for (int q = 1; q < 16; ++q) {
                const float2 t = cmul(wi, A.tw[(q - 1) * 256 + tid]);
                cx_set(twC[q], t.x, t.y);
            }
        }
        __syncthreads();  // s_b
        float sbx[16], sby[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float2 t = s_b[p];
            sbx[p] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t.x)));
            sby[p] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t.y)));
        }
        // (the rows are fetched after the twiddle set-up: loaded before it, they and its temporaries overflow the register file)
        fetch_x(g0);
        // the code-spectrum rows of every component stay in registers (packed) for all cells
        uint32_t cv[NCOMP][16];
#pragma unroll
        for (int comp = 0; comp < NCOMP; ++comp) {
            fetch16(Cs + (long)comp * L + (long)k1 * S, cv[comp], std::integral_constant<bool, kRowsNtC>{});
        }
        PH_MARK(16);  // workgroup prologue: twiddles, code rows issued
        for (int g = g0; g < g1; ++g) {
            uint32_t r0[ILV_OK ? 16 : 1];  // packed outputs of component 0, held for the interleaved store
#pragma unroll
            for (int comp = 0; comp < NCOMP; ++comp) {
                // ---- phase 1a: products, radix 16 over bh, to the wave's region.  (Round 4: every exchange is spread over the
                // butterfly layer that produces / consumes it -- the four outputs of a layer-2 group are written while the next
                // group is computed, reads are issued in the order the layer-1 groups need them -- instead of 16 writes and 16
                // reads back to back: a burst from all four waves queues on the LDS store path at ~40 cycles per write against
                // ~20 spread out, tools/phases.py.)
                C y[16], a[4][4], o[4];
#ifndef BDS_ROWS_DOT2_BUILTIN  // (-DBDS_ROWS_DOT2_BUILTIN: the compiler builtin instead; rows 1.82 vs 1.78 ms per 201 cells)
                // The products with the three-operand v_dot2_f32_f16 (addend 0 inline) in inline assembly: the builtin compiles to
                // the accumulating v_dot2c_f32_f16 behind a v_mov 0 per result (64 moves per cell).  The compiler does not know
                // these are dot instructions, so the hazard it would pad -- 3 wait states between a dot's write and another
                // vector instruction's read -- is closed by hand: four products per block, s_nop 2 at its end.
#pragma unroll
                for (int q = 0; q < 16; q += 4) {
                    uint32_t xs[4], xc[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        xs[i] = __builtin_amdgcn_alignbit(xn[q + i], xn[q + i], 16);  // (xi, xr)
                        xc[i] = xn[q + i] ^ 0x80000000u;                               // (xr, -xi)
                    }
                    float re[4], im[4];
                    asm volatile(
                        "v_dot2_f32_f16 %0, %8, %16, 0\n v_dot2_f32_f16 %4, %12, %16, 0\n"
                        "v_dot2_f32_f16 %1, %9, %17, 0\n v_dot2_f32_f16 %5, %13, %17, 0\n"
                        "v_dot2_f32_f16 %2, %10, %18, 0\n v_dot2_f32_f16 %6, %14, %18, 0\n"
                        "v_dot2_f32_f16 %3, %11, %19, 0\n v_dot2_f32_f16 %7, %15, %19, 0\n s_nop 2"
                        : "=&v"(re[0]), "=&v"(re[1]), "=&v"(re[2]), "=&v"(re[3]), "=&v"(im[0]), "=&v"(im[1]), "=&v"(im[2]), "=&v"(im[3])
                        : "v"(xc[0]), "v"(xc[1]), "v"(xc[2]), "v"(xc[3]), "v"(xs[0]), "v"(xs[1]), "v"(xs[2]), "v"(xs[3]),
                          "v"(cv[comp][q]), "v"(cv[comp][q + 1]), "v"(cv[comp][q + 2]), "v"(cv[comp][q + 3]));
#pragma unroll
                    for (int i = 0; i < 4; ++i) cx_set(y[q + i], re[i], im[i]);
                }
#else
#pragma unroll
                for (int q = 0; q < 16; ++q) {  // ((xi, xr) is re-formed per component: one v_alignbit against 16 registers held across both)
                    const float2 t = cmul_h(xn[q], __builtin_amdgcn_alignbit(xn[q], xn[q], 16), cv[comp][q]);
                    cx_set(y[q], t.x, t.y);
                }
#endif
                // the last component's products are the last readers of xn: the next cell's row is fetched into the same
                // registers while the transform and the stores run
                if (comp == NCOMP - 1 && g + 1 < g1) fetch_x(g + 1);
#pragma unroll
                for (int n2 = 0; n2 < 4; ++n2) cx_bfly16_l1<false>(y, (const C *)nullptr, n2, a[n2]);
#define BDS_WR_L2(K1, DST, STEP, ODD)                       \
    cx_bfly16_l2<K1>(a, o);                                 \
    __builtin_amdgcn_sched_barrier(0);                      \
    (DST)[(STEP) * (K1) + (ODD) * ((K1) & 1)] = o[0];       \
    (DST)[(STEP) * ((K1) + 4) + (ODD) * ((K1) & 1)] = o[1]; \
    (DST)[(STEP) * ((K1) + 8) + (ODD) * ((K1) & 1)] = o[2]; \
    (DST)[(STEP) * ((K1) + 12) + (ODD) * ((K1) & 1)] = o[3]; \
    __builtin_amdgcn_sched_barrier(0)
                BDS_WR_L2(0, wr1, 64, 0);
                BDS_WR_L2(1, wr1, 64, 0);
                BDS_WR_L2(2, wr1, 64, 0);
                BDS_WR_L2(3, wr1, 64, 0);
                wave_sync();
                // ---- phase 1b: radix 16 over bl (rotated start), to the exchange buffer
#pragma unroll
                for (int n2 = 0; n2 < 4; ++n2) {
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) y[n2 + 4 * mm] = rd1[4 * ((n2 + 4 * mm + bl) & 15)];
                }
                wave_sync();
                __builtin_amdgcn_sched_barrier(0);
                PH_MARK(8 * comp + 0);  // products, phase 1a, exchange issued
#pragma unroll
                for (int n2 = 0; n2 < 4; ++n2) cx_bfly16_l1<true>(y, twB, n2, a[n2]);  // twiddle w256^(bl u) on the inputs (up to the factor w256^(u u))
                PH_MARK(8 * comp + 2);  // phase 1b layer 1 (waits for its inputs group by group)
                if (comp > 0 || g > g0) BDS_SYNC();  // every thread is through with the exchange buffer (previous transform)
                PH_MARK(8 * comp + 3);  // barrier
                BDS_WR_L2(0, wrx, 16 * XS, 2);
                BDS_WR_L2(1, wrx, 16 * XS, 2);
                BDS_WR_L2(2, wrx, 16 * XS, 2);
                BDS_WR_L2(3, wrx, 16 * XS, 2);
#undef BDS_WR_L2
                PH_MARK(8 * comp + 4);  // phase 1b layer 2 + exchange writes
                BDS_SYNC();
                PH_MARK(8 * comp + 5);  // barrier
                // ---- phase 2: twiddle (wi folded in), radix 16 over q', uniform factor, store
                lds_read16(y, rd2a);  // bds_lds.h: sixteen ds_read_b64 (the compiler's ds_read2_b64 pairs take twice the LDS cycles)
                __builtin_amdgcn_sched_barrier(0);
                PH_MARK(8 * comp + 6);  // exchange reads issued
#pragma unroll
                for (int n2 = 0; n2 < 4; ++n2) cx_bfly16_l1<true, true>(y, twC, n2, a[n2]);
                auto emit = [&](int p, C yv) {
                    const float2 t = cx_f2(cx_mul_uniform(yv, sbx[p], sby[p]));
                    const uint32_t h = f2_to_h2(t);
#ifdef BDS_EXP_ROWS_NOSTORE
                    if (t.x == 1.2345f)
#endif
                    {
                        if constexpr (ILV) {
                            if (comp == 0) {
                                r0[ILV_OK ? p : 0] = h;
                            } else {
                                typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                                u2v *dst2 = (u2v *)A.Bw + (long)g * L + (long)k1 * S + tid + 256 * p;
                                const u2v val = {r0[ILV_OK ? p : 0], h};
                                if constexpr (kRowsNtSt)
                                    __builtin_nontemporal_store(val, dst2);
                                else
                                    *dst2 = val;
                            }
                        } else {
                            ST *dst = (ST *)A.Bw + ((long)g * NCOMP + comp) * L + (long)k1 * S + tid;
                            *reinterpret_cast<uint32_t *>(dst + 256 * p) = h;
                        }
                    }
                };
#define BDS_ST_L2(K1)                                     \
    cx_bfly16_l2<K1>(a, o);                               \
    emit((K1), o[0]), emit((K1) + 4, o[1]), emit((K1) + 8, o[2]), emit((K1) + 12, o[3])
                BDS_ST_L2(0);
                BDS_ST_L2(1);
                BDS_ST_L2(2);
                BDS_ST_L2(3);
#undef BDS_ST_L2
                PH_MARK(8 * comp + 7);  // phase 2 arithmetic, stores issued
            }
        }
        if (vb + (int)gridDim.x < A.nvb) BDS_SYNC();
    }
    PH_FLUSH(32, 17);
    clkp.finish(tid);
}

}  // namespace bds
