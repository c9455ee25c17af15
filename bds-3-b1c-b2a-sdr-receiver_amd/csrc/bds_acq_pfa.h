// N-point search pair for the B1C grid at N = 1 987 500 = 53 x 12 x 3125 (round 6; tools/proto_pfa53.py is the NumPy model).
//
// The reference correlates circularly over N = len10PlusXms samples (B1C/acquisition.m:135-136, 198-212).  The pair of
// bds_acq_wrows.h / bds_acq_wcols.h transforms L = 3 145 728 = 1.583 N points per cell (zero-padded linear correlation); this
// pair transforms N.  53, 12 and 3125 are pairwise coprime, so the N-point transform is a 3-D transform WITHOUT twiddles between
// the dimensions (Good-Thomas): spectrum index k <-> (k1, k2, k3) = (k mod 53, k mod 12, k mod 3125), lag index
// t = (t1 N/53 + t2 N/12 + t3 N/3125) mod N, and the Doppler bins -- acqStep N / fs = 1 at cfg3 -- are rotations of ONE signal
// spectrum by the bin index in each dimension: fft(carr_b x)[k] = fft(carr_0 x)[k - b].
//
//   row pass     k_pfa_rows   workgroup = two rows (k1 = 2 mp, 2 mp + 1; one k2) x both components x a run of cells of one PRN:
//                             spectrum product (v_dot2_f32_f16 on fp16-stored spectra, code rows held in registers across the
//                             cells) + inverse 3125-point transform over k3 as 25 x 5 x 25 on 125 threads per row (packed fp32
//                             butterflies, two LDS exchanges), fp16 result to the inter-pass buffer.  A row lives in one 32-lane
//                             HALF of four waves, its partner row in the other half: the two rows' results meet through
//                             v_permlane32_swap and leave as 16-byte pieces, 512 contiguous bytes per half-wave.
//   column pass  k_pfa_cols   wave = 4 lags t3 x all (t1, t2) x both components: the 53-point transform over k1 on the MATRIX pipe --
//                             the inter-pass buffer holds fp16 values, which v_mfma_f32_16x16x32_f16 multiplies exactly into fp32;
//                             the DFT coefficients are split hi + lo in fp16 (error ~1e-7, tools/proto_pfa53.py) --, the 12-point
//                             transform over k2 per lane on the accumulators (a lane holds all 12 k2 of one output: data is the
//                             A operand, so the MFMA's rows are (t3, k2) and its columns the outputs), |y|^2, the sieve protocol
//                             of bds_acq_wcols.h.
//
// Inter-pass buffer of a cell: [mp 27][k2 12][t3 3125][component 2][row of the pair 2] fp16 complex = 16 bytes per (mp, k2, t3);
// a lane's A fragment (k1 = 4 mg .. 4 mg + 3 of one (k2, t3), one component) is the halves of two such pieces.
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <type_traits>

#include "bds_acq_wcols.h"  // Extra, wc_pack, wave_max_f32; bds_fft_pk.h, bds_lds.h

#ifdef PFA_EXP_R_NOBAR
#define PFA_RSYNC() __builtin_amdgcn_s_waitcnt(0)
#else
#define PFA_RSYNC() __syncthreads()
#endif

namespace bds {
namespace pfa {

constexpr int K1 = 53, K2 = 12, K3 = 3125;
constexpr long NP = (long)K1 * K2 * K3;  // 1 987 500
constexpr int MP = 27;                   // row pairs (54 rows: one zero row)
constexpr int NB = 7;                    // output blocks of 16 (106 real outputs -> 112)
// inter-pass buffer of a cell: [tile of 16 lags t3 (196)][row pair mp (27)][k2 (12)][lag in the tile (16)][component 2][row of the pair 2],
// fp16 complex: a column workgroup's item (16 lags, all 324 (mp, k2)) is ONE contiguous 83 KB block, a (mp, k2, lag) piece 16 bytes.
// (Round 6 first had [mp][k2][t3 3125]: 50 KB contiguous per row workgroup, but 324 pieces of 64 bytes 50 KB apart per column wave --
//  columns 1.15 ms per 201 cells against 1.04 with the tiles; the row pass does not notice its 256-byte runs.)
constexpr int kTileLags = 16, kTiles = (K3 + kTileLags - 1) / kTileLags;
constexpr size_t kCellElems = (size_t)kTiles * MP * K2 * kTileLags * 4;  // 4-byte (fp16 complex) elements of a cell
__host__ __device__ constexpr size_t bw_piece(int mp, int k2, int t3) {  // element index of the 4-element piece of (mp, k2, t3) in its cell
    return (((size_t)(t3 / kTileLags) * MP + mp) * K2 + k2) * (kTileLags * 4) + (size_t)(t3 % kTileLags) * 4;
}  // 4-byte (fp16 complex) elements of a cell in the inter-pass buffer
constexpr int kRowsThreads = 256, kColsThreads = 256;
constexpr int kCoefFrags = NB * 4 * 2;  // B fragments (16 bytes per lane): [nb][ins][hi / lo]
constexpr size_t kCoefBytes = (size_t)kCoefFrags * 64 * 16;
constexpr size_t kColsLds = kCoefBytes + (size_t)NB * kColsThreads * 4;  // + the lanes' maxima per output block (the values' pass visits flagged blocks only)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// lag of output (t1, t2, t3): the Ruritanian map
__host__ __device__ inline long lag_of(int t1, int t2, int t3) {
    return ((long)t1 * (NP / K1) + (long)t2 * (NP / K2) + (long)t3 * (NP / K3)) % NP;
}

// ---- 5- and 25-point inverse transforms on packed fp32 pairs -------------------------------------------------------------
__device__ __forceinline__ void pk_radix5(v2f &x0, v2f &x1, v2f &x2, v2f &x3, v2f &x4) {
    constexpr float c1 = 0.30901699437494745f, c2 = -0.80901699437494734f, s1 = 0.95105651629515353f, s2 = 0.58778525229247314f;
    const v2f t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    const v2f m1 = x0 + c1 * t1 + c2 * t2;
    const v2f m2 = x0 + c2 * t1 + c1 * t2;
    const v2f n1 = s1 * t3 + s2 * t4;
    const v2f n2 = s2 * t3 - s1 * t4;
    x0 = x0 + t1 + t2;
    x1 = pk_addj(m1, n1);
    x4 = pk_subj(m1, n1);
    x2 = pk_addj(m2, n2);
    x3 = pk_subj(m2, n2);
}

// W25^k, k = q0 p1 for q0, p1 in 1..4
__device__ __forceinline__ v2f w25(int k) {
    constexpr float c[17] = {1.f, 0.9685831611286311f, 0.8763066800438636f, 0.7289686274214116f, 0.5358267949789965f, 0.30901699437494745f,
                             0.06279051952931353f, -0.1873813145857246f, -0.4257792915650727f, -0.6374239897486897f, -0.8090169943749473f,
                             -0.9297764858882513f, -0.9921147013144778f, -0.9921147013144779f, -0.9297764858882515f, -0.8090169943749478f,
                             -0.6374239897486895f};
    constexpr float s[17] = {0.f, 0.2486898871648548f, 0.4817536741017153f, 0.6845471059286886f, 0.8443279255020151f, 0.9510565162951535f,
                             0.9980267284282716f, 0.9822872507286887f, 0.9048270524660195f, 0.7705132427757893f, 0.5877852522924732f,
                             0.36812455268467814f, 0.12533323356430454f, -0.12533323356430429f, -0.3681245526846779f, -0.5877852522924727f,
                             -0.7705132427757894f};
    return (v2f){c[k], s[k]};
}

// x[q0 + 5 q1] -> slot p0 + 5 p1 holds Y[5 p0 + p1] = sum_q x[q] W25^(q (5 p0 + p1))
__device__ __forceinline__ void pk_radix25(v2f (&x)[25]) {
#pragma unroll
    for (int q0 = 0; q0 < 5; ++q0) pk_radix5(x[q0], x[q0 + 5], x[q0 + 10], x[q0 + 15], x[q0 + 20]);
#pragma unroll
    for (int q0 = 1; q0 < 5; ++q0)
#pragma unroll
        for (int p1 = 1; p1 < 5; ++p1) x[q0 + 5 * p1] = pk_cmul_k(x[q0 + 5 * p1], w25(q0 * p1));
#pragma unroll
    for (int p1 = 0; p1 < 5; ++p1) pk_radix5(x[5 * p1], x[5 * p1 + 1], x[5 * p1 + 2], x[5 * p1 + 3], x[5 * p1 + 4]);
}
__host__ __device__ constexpr int slot25_index(int s) { return 5 * (s % 5) + s / 5; }  // output index held by slot s

__device__ __forceinline__ v2f unit(int num, int den) {  // exp(+2 pi j num / den), num reduced by the caller
    float sn, cs;
    sincospif(2.0f * (float)num / (float)den, &sn, &cs);
    return (v2f){cs, sn};
}

// LDS reads of the row pass as single ds_read_b64 (bds_lds.h: the ds_read2_b64 the compiler fuses two reads into is served in groups of
// 16 lanes at half the rate): y[k] = *(a + BASE + k STEP), byte offsets, one wait per batch
template <int BASE, int STEP>
__device__ __forceinline__ void lds_read5(v2f (&y)[5], unsigned a) {
    asm volatile(
        "ds_read_b64 %0, %5 offset:%6\n ds_read_b64 %1, %5 offset:%6+%7\n ds_read_b64 %2, %5 offset:%6+2*%7\n ds_read_b64 %3, %5 offset:%6+3*%7\n"
        "ds_read_b64 %4, %5 offset:%6+4*%7\n s_waitcnt lgkmcnt(0)"
        : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4])
        : "v"(a), "n"(BASE), "n"(STEP)
        : "memory");
}
template <int STEP>
__device__ __forceinline__ void lds_read25(v2f (&y)[25], unsigned a) {
    asm volatile(
        "ds_read_b64 %0, %25\n ds_read_b64 %1, %25 offset:%26\n ds_read_b64 %2, %25 offset:2*%26\n ds_read_b64 %3, %25 offset:3*%26\n ds_read_b64 %4, %25 offset:4*%26\n"
        "ds_read_b64 %5, %25 offset:5*%26\n ds_read_b64 %6, %25 offset:6*%26\n ds_read_b64 %7, %25 offset:7*%26\n ds_read_b64 %8, %25 offset:8*%26\n ds_read_b64 %9, %25 offset:9*%26\n"
        "ds_read_b64 %10, %25 offset:10*%26\n ds_read_b64 %11, %25 offset:11*%26\n ds_read_b64 %12, %25 offset:12*%26\n ds_read_b64 %13, %25 offset:13*%26\n ds_read_b64 %14, %25 offset:14*%26\n"
        "ds_read_b64 %15, %25 offset:15*%26\n ds_read_b64 %16, %25 offset:16*%26\n ds_read_b64 %17, %25 offset:17*%26\n ds_read_b64 %18, %25 offset:18*%26\n ds_read_b64 %19, %25 offset:19*%26\n"
        "ds_read_b64 %20, %25 offset:20*%26\n ds_read_b64 %21, %25 offset:21*%26\n ds_read_b64 %22, %25 offset:22*%26\n ds_read_b64 %23, %25 offset:23*%26\n ds_read_b64 %24, %25 offset:24*%26\n"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]), "=&v"(y[6]), "=&v"(y[7]), "=&v"(y[8]), "=&v"(y[9]), "=&v"(y[10]), "=&v"(y[11]),
          "=&v"(y[12]), "=&v"(y[13]), "=&v"(y[14]), "=&v"(y[15]), "=&v"(y[16]), "=&v"(y[17]), "=&v"(y[18]), "=&v"(y[19]), "=&v"(y[20]), "=&v"(y[21]), "=&v"(y[22]),
          "=&v"(y[23]), "=&v"(y[24])
        : "v"(a), "n"(STEP)
        : "memory");
}

// ---- row pass ----------------------------------------------------------------------------------------------------------------
struct RowsArgs {
    const uint32_t *Xs;  // signal spectrum of bin 0, CRT layout, every row doubled: [53][12][6250] fp16 complex (re lo, im hi)
    const uint32_t *Cs;  // conjugated, scaled code spectra: [prn slot][component][53][12][3125]
    uint32_t *Bw;        // inter-pass buffer [cell][27][12][3125][2][2]
    const int *bin;      // per cell: Doppler bin (= rotation)
    const long *cs;      // per cell: element offset of the PRN's spectra in Cs
    int ncells;          // cells of the launch
    int gc;              // cells a workgroup walks (all of one PRN)
    int shift;           // spectrum bins per Doppler bin (acqStep N / fs)
};

template <int NC>
__global__ __launch_bounds__(kRowsThreads, 2) void k_pfa_rows(RowsArgs A) {
    extern __shared__ __align__(16) unsigned char pfa_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
    const int j = wave * 32 + (lane & 31);  // thread of the row: 0..127, 125 of them work
    const bool live = j < 125;
    const int jj = live ? j : 124;
    const int rp = blockIdx.x % (MP * K2), chunk = blockIdx.x / (MP * K2);
    const int mp = rp / K2, k2 = rp % K2;
    const int k1 = 2 * mp + half;
    const bool row_ok = k1 < K1;
    const int k1c = row_ok ? k1 : K1 - 1;
    float2 *region = reinterpret_cast<float2 *>(pfa_lds) + (size_t)half * 3136;  // 3125 elements per row (+ pad)
    const int c0 = chunk * A.gc, c1 = min(A.ncells, c0 + A.gc);
    if (c0 >= c1) return;

    // code rows of this workgroup's PRN (zero for the pad row: its outputs are zeros)
    uint32_t cv[NC][25];
    {
        const uint32_t *crow = A.Cs + A.cs[c0] + ((size_t)k1c * K2 + k2) * K3;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 25; ++q) cv[c][q] = row_ok ? crow[(size_t)c * K1 * K2 * K3 + jj + 125 * q] : 0u;
    }
    // stage twiddles: W3125^(j p) after stage 1, W125^(i u) after stage 2 (thread t = i + 25 pg)
    // (W3125^(j p), p = 5 p0 + p1, as the product of two factors from 4 + 4 registers: all 24 of them held per thread -- 48 registers --
    //  spill 43 dwords at two waves per SIMD; the 16 extra complex products are 5 % of the row pass's vector instructions)
    v2f tw1a[5], tw1b[5], tw2[5];
#pragma unroll
    for (int p = 1; p < 5; ++p) tw1a[p] = unit((jj * p) % K3, K3), tw1b[p] = unit((jj * 5 * p) % K3, K3);
    // stage-2 thread t = 5 i + pg: its five reads / in-place writes sit at 25 (i + 25 r) + 5 pg + c = 5 t + (625 r + c) -- stride 5 over
    // the lanes, conflict-free in the 32-lane halves of a ds_read_b64 and the 16-lane groups of a ds_write_b64 (t = i + 25 pg, the
    // first version, was 2-way in every half-wave: SQ_LDS_BANK_CONFLICT 28 % of the LDS cycles, profiles/r06_b1c_pmc_first.txt)
    const int si = jj / 5, spg = jj % 5;
    const unsigned region_b = lds_offset(region);
#pragma unroll
    for (int u = 1; u < 5; ++u) tw2[u] = unit((si * u) % 125, 125);

    const __amdgpu_buffer_rsrc_t xs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)A.Xs, 0, K1 * K2 * 2 * K3 * 4, 0x00020000);
    // The next cell's Doppler bin is requested right behind this cell's signal loads and taken into a scalar in front of this cell's
    // stores: a vector load at the head of the loop would be waited for with vmcnt(0), i.e. together with the stores just issued.
    int bin_cur = A.bin[c0];
    for (int cell = c0; cell < c1; ++cell) {
        const int s = bin_cur * A.shift;
        const int k1s = ((k1c - s) % K1 + K1) % K1, k2s = ((k2 - s) % K2 + K2) % K2, o3 = (K3 - s % K3) % K3;
        // (buffer loads / stores: one descriptor, one lane offset, the 25 strides as immediates -- per-access 64-bit address arithmetic was
        //  five vector instructions per load: 8 % of the cell loop)
        const int xoff = ((k1s * K2 + k2s) * (2 * K3) + o3 + jj) * 4;  // < 2^25 bytes
        uint32_t xn[25];
#ifdef PFA_EXP_R_NOLOAD
#pragma unroll
        for (int q = 0; q < 25; ++q) xn[q] = 0x3c003800u + q + cell + (uint32_t)xoff;
#else
#pragma unroll
        for (int q = 0; q < 25; ++q) xn[q] = __builtin_amdgcn_raw_buffer_load_b32(xs_rsrc, xoff, 500 * q, 0);
#endif
        int bin_next = A.bin[min(cell + 1, c1 - 1)];
        uint32_t outp[NC][25];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            v2f x[25];
            // X conj(C): (xr, -xi).(cr', ci') and (xi, xr).(cr', ci') with C' = conj(C) stored.  The three-operand v_dot2_f32_f16 with an inline
            // zero addend, five products per asm block with the dot -> VALU hazard closed by hand (s_nop 2), as bds_acq_wrows.h: the
            // builtin compiles to the accumulating v_dot2c behind a v_mov 0 per result (50 moves per cell).  -DPFA_ROWS_DOT2_BUILTIN: the builtin.
#ifndef PFA_ROWS_DOT2_BUILTIN
#pragma unroll
            for (int q = 0; q < 25; q += 5) {
                uint32_t xs[5], xc[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) xs[i] = __builtin_amdgcn_alignbit(xn[q + i], xn[q + i], 16), xc[i] = xn[q + i];
                float re[5], im[5];
                // (neg_hi on the signal word: (xr, -xi) without an xor per point -- the dot instructions take neg_lo / neg_hi, not op_sel,
                //  so the swapped word of the imaginary part is still prepared: tools/probe/dot2_mods.hip)
                asm volatile(
                    "v_dot2_f32_f16 %0, %10, %20, 0 neg_hi:[1,0,0]\n v_dot2_f32_f16 %5, %15, %20, 0\n"
                    "v_dot2_f32_f16 %1, %11, %21, 0 neg_hi:[1,0,0]\n v_dot2_f32_f16 %6, %16, %21, 0\n"
                    "v_dot2_f32_f16 %2, %12, %22, 0 neg_hi:[1,0,0]\n v_dot2_f32_f16 %7, %17, %22, 0\n"
                    "v_dot2_f32_f16 %3, %13, %23, 0 neg_hi:[1,0,0]\n v_dot2_f32_f16 %8, %18, %23, 0\n"
                    "v_dot2_f32_f16 %4, %14, %24, 0 neg_hi:[1,0,0]\n v_dot2_f32_f16 %9, %19, %24, 0\n s_nop 2"
                    : "=&v"(re[0]), "=&v"(re[1]), "=&v"(re[2]), "=&v"(re[3]), "=&v"(re[4]), "=&v"(im[0]), "=&v"(im[1]), "=&v"(im[2]), "=&v"(im[3]), "=&v"(im[4])
                    : "v"(xc[0]), "v"(xc[1]), "v"(xc[2]), "v"(xc[3]), "v"(xc[4]), "v"(xs[0]), "v"(xs[1]), "v"(xs[2]), "v"(xs[3]), "v"(xs[4]),
                      "v"(cv[c][q]), "v"(cv[c][q + 1]), "v"(cv[c][q + 2]), "v"(cv[c][q + 3]), "v"(cv[c][q + 4]));
#pragma unroll
                for (int i = 0; i < 5; ++i) x[q + i] = (v2f){re[i], im[i]};
            }
#else
#pragma unroll
            for (int q = 0; q < 25; ++q) {
                const uint32_t xs = __builtin_amdgcn_alignbit(xn[q], xn[q], 16), xc = xn[q] ^ 0x80000000u;
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 hc = __builtin_bit_cast(h2, cv[c][q]);
                x[q] = (v2f){__builtin_amdgcn_fdot2(__builtin_bit_cast(h2, xc), hc, 0.f, false),
                             __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, xs), hc, 0.f, false)};
            }
#endif
            // stage 1: 25 points over q (k3 = j + 125 q) -> p, twiddle W3125^(j p), a[j][p] at 25 j + p
            pk_radix25(x);
            if (c > 0) PFA_RSYNC();  // the previous component's stage-3 reads of the region are done
            if (live) {
#pragma unroll
                for (int sl = 0; sl < 25; ++sl) {
                    const int p = slot25_index(sl);  // slot p0 + 5 p1 holds output 5 p0 + p1
                    v2f v = x[sl];
                    if (sl / 5) v = pk_cmul(v, tw1a[sl / 5]);
                    if (sl % 5) v = pk_cmul(v, tw1b[sl % 5]);
                    region[25 * j + p] = to_f2(v);
                }
            }
            PFA_RSYNC();
            // stage 2: thread (i, pg): for c5 = 0..4: 5 points over r of a[i + 25 r][5 pg + c5] -> u, twiddle W125^(i u), in place
            if (live) {
#pragma unroll
                for (int c5 = 0; c5 < 5; ++c5) {
                    v2f z[5];
                    // a[i + 25 r][5 pg + c5], r = 0..4: 625 elements apart
                    lds_read5<0, 625 * 8>(z, region_b + (unsigned)(25 * si + 5 * spg + c5) * 8u);
                    pk_radix5(z[0], z[1], z[2], z[3], z[4]);
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const v2f v = u ? pk_cmul(z[u], tw2[u]) : z[u];
                        region[25 * (si + 25 * u) + 5 * spg + c5] = to_f2(v);
                    }
                }
            }
            PFA_RSYNC();
            // stage 3: thread t' = p + 25 u: 25 points over i of b[p][i][u] at 25 (i + 25 u) + p -> t'': X[t' + 125 t'']
            lds_read25<25 * 8>(x, region_b + (unsigned)(625 * (jj / 25) + jj % 25) * 8u);  // b[p][i][u] at 25 (i + 25 u) + p, i = 0..24
            pk_radix25(x);
#pragma unroll
            for (int sl = 0; sl < 25; ++sl) {
                const int tq = slot25_index(sl);
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                outp[c][tq] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x[sl], h2));  // v_cvt_pk_f16_f32: round to nearest even
            }
        }
        // the two rows of the pair meet: after the swap half 0 holds (row 0, row 1) of t'' = e, half 1 of t'' = e + 1
        // (descriptor at the piece of lag 0: a lag's offset is tile * 82 944 + (lag in the tile) * 16 bytes)
        const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(A.Bw + (size_t)cell * kCellElems + bw_piece(mp, k2, 0)), 0,
                                                                                   (unsigned)((kCellElems - bw_piece(mp, k2, 0)) * 4), 0x00020000);
        asm volatile("" : "+v"(bin_next));  // (the bin's wait stands here: every load long back, no store in flight yet)
        bin_cur = __builtin_amdgcn_readfirstlane(bin_next);
        typedef int v4i __attribute__((ext_vector_type(4)));
#ifdef PFA_EXP_R_SAMECELL  // (timing experiment: every cell's stores into the first cell's 16 MB -- the Infinity Cache takes them)
        const unsigned long long dst_base = (unsigned long long)(A.Bw + bw_piece(mp, k2, 0));
#else
        const unsigned long long dst_base = (unsigned long long)(A.Bw + (size_t)cell * kCellElems + bw_piece(mp, k2, 0));
#endif
        const v4i dst_words = {(int)(unsigned)dst_base, (int)((unsigned)(dst_base >> 32) & 0xffffu), (int)(unsigned)((kCellElems - bw_piece(mp, k2, 0)) * 4), 0x00020000};
        (void)dst_rsrc;
#pragma unroll
        for (int e = 0; e < 25; e += 2) {
            uint32_t P[2], Q[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t pe = c < NC ? outp[c < NC ? c : 0][e] : 0u, qo = (c < NC && e + 1 < 25) ? outp[c < NC ? c : 0][e + 1 < 25 ? e + 1 : e] : 0u;
                // v_permlane32_swap: lanes 32-63 of the first <-> lanes 0-31 of the second
                const auto r = __builtin_amdgcn_permlane32_swap(pe, qo, false, false);
                P[c] = r[0], Q[c] = r[1];
            }
            const int tq = e + half;
            if (live && tq < 25) {  // lag t3 = j + 125 tq
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                const int t3s = j + 125 * tq;
#ifdef PFA_EXP_R_NOSTORE
                asm volatile("" ::"v"(P[0]), "v"(Q[0]), "v"(P[1]), "v"(Q[1]), "v"(t3s));
#else
#ifdef PFA_ROWS_STORE_BUILTIN
                __builtin_amdgcn_raw_buffer_store_b128((u4){P[0], Q[0], P[1], Q[1]}, dst_rsrc, (int)(bw_piece(0, 0, t3s) * 4), 0, 0);
#else
                // Issued by hand: loads and stores share the vmcnt counter and return out of order relative to each other, so with stores
                // the compiler knows of in flight its first wait for a load of the NEXT cell becomes vmcnt(0) -- the whole latency of this
                // cell's 13 stores in front of every cell (the row pass with its stores compiled out ran 17 % faster).  Stores it does not
                // know of only make its counted waits longer, never shorter: of the k + S operations a wait lets return, at most S are stores.
                // Nothing in this kernel reads the buffer; the end of the kernel makes the stores visible.
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n s_nop 0" ::"v"((u4){P[0], Q[0], P[1], Q[1]}), "v"((int)(bw_piece(0, 0, t3s) * 4)), "s"(dst_words)
                             : "memory");
#endif
#endif
            }
        }
        PFA_RSYNC();  // region free for the next cell
    }
}

// ---- column pass -------------------------------------------------------------------------------------------------------------
// B operand of the 53-point stage in fragment order: frag[(nb * 4 + ins) * 2 + part][lane] = 8 halves
//   coef[kappa = 32 ins + 8 (lane >> 4) + e][o = 16 nb + (lane & 15)],  kappa = 2 k1 + ri_in, o = 2 t1 + ri_out,
//   y_re = sum c x_re - s x_im, y_im = sum s x_re + c x_im, c + j s = exp(+2 pi j k1 t1 / 53); part 0 = fp16(coef), part 1 = fp16(coef - hi)
inline void make_coef_frags(uint16_t *out /* kCoefBytes / 2 halves */) {
    auto f2h = [](float f) { return __half_as_ushort(__float2half_rn(f)); };
    auto h2f = [](uint16_t h) { return __half2float(__ushort_as_half(h)); };
    for (int nb = 0; nb < NB; ++nb)
        for (int ins = 0; ins < 4; ++ins)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int kap = 32 * ins + 8 * (lane >> 4) + e, o = 16 * nb + (lane & 15);
                    const int k1 = kap >> 1, ri = kap & 1, t1 = o >> 1, ro = o & 1;
                    double v = 0.0;
                    if (k1 < K1 && t1 < K1) {
                        const double ang = 2.0 * M_PI * (double)((k1 * t1) % K1) / K1;
                        const double c = cos(ang), s = sin(ang);
                        v = ro == 0 ? (ri == 0 ? c : -s) : (ri == 0 ? s : c);
                    }
                    const uint16_t hi = f2h((float)v);
                    const uint16_t lo = f2h((float)(v - (double)h2f(hi)));
                    out[((((size_t)nb * 4 + ins) * 2 + 0) * 64 + lane) * 8 + e] = hi;
                    out[((((size_t)nb * 4 + ins) * 2 + 1) * 64 + lane) * 8 + e] = lo;
                }
}

#ifndef PFA_COLS_BOUND_PARTS
#define PFA_COLS_BOUND_PARTS 1
#endif
constexpr int kBoundParts = PFA_COLS_BOUND_PARTS;  // coefficient parts of the column pass's bound pass (1: hi only; 2: as the values' pass)

struct ColsArgs {
    const uint32_t *Bw;           // inter-pass buffer of the launch's cells
    const uint4 *coef;            // make_coef_frags
    int ncells;                   // cells of the launch
    float w0, w1;                 // magnitude weights (storage scales undone)
    unsigned long long *cellmax;  // [run-wide cell]: (value bits << 32) | ~lag, by atomic max      -- the protocol of bds_acq_wcols.h --
    float *lb;                    // [(run-wide cell) / lb_div]: running lower bound of that PRN's sieve maximum
    int lb_div;
    Extra *extra;                 // candidate list
    int *extra_count;
    int extra_cap;
    int cell0;                    // run-wide index of cell 0 of this launch
    float keep;                   // 1 - tolerance of the sieve
    int qchunk;                   // blocks of 16 lags of a cell that follow each other in the work list
    unsigned long long *stats;    // optional (probe): [0] wave items, [1] of them through the values' pass, [2] through the exhaustive pass, [3] output blocks the values' pass visited
    float *dbg;                   // optional: |y_d|^2, |y_p|^2 of one (cell, t3 group): [2][53][12][4]
    int dbg_cell, dbg_group;
};

// real 12-point transform F[t] = sum_k v[k] exp(+2 pi j k t / 12) of a real sequence: P = re F (t = 0..6), Q = im F (t = 1..5)
__device__ __forceinline__ void real_dft6(float a0, float a1, float a2, float a3, float a4, float a5, float (&re)[4], float (&im)[4]) {
    constexpr float h3 = 0.86602540378443865f;
    const float s0 = a0 + a3, d0 = a0 - a3, s1 = a1 + a4, d1 = a1 - a4, s2 = a2 + a5, d2 = a2 - a5;
    const float s12 = s1 + s2, d12 = d1 - d2;
    re[0] = s0 + s12, im[0] = 0.f;
    re[2] = fmaf(-0.5f, s12, s0), im[2] = h3 * (s1 - s2);
    re[1] = fmaf(0.5f, d12, d0), im[1] = h3 * (d1 + d2);
    re[3] = d0 - d12, im[3] = 0.f;
}
__device__ __forceinline__ void real_dft12(const float (&v)[12], float (&P)[7], float (&Q)[7]) {
    float er[4], ei[4], orr[4], oi[4];
    real_dft6(v[0], v[2], v[4], v[6], v[8], v[10], er, ei);
    real_dft6(v[1], v[3], v[5], v[7], v[9], v[11], orr, oi);
    constexpr float c1 = 0.86602540378443865f, s1 = 0.5f;  // W12 = exp(j pi / 6)
    // F[t] = E[t mod 6] + W12^t O[t mod 6];  E[6 - t] = conj E[t]
    P[0] = er[0] + orr[0], Q[0] = 0.f;
    P[6] = er[0] - orr[0], Q[6] = 0.f;
    P[3] = er[3], Q[3] = orr[3];  // W12^3 = j, E[3], O[3] real
    P[1] = er[1] + c1 * orr[1] - s1 * oi[1], Q[1] = ei[1] + c1 * oi[1] + s1 * orr[1];
    P[5] = er[1] - c1 * orr[1] + s1 * oi[1], Q[5] = -ei[1] + c1 * oi[1] + s1 * orr[1];  // E[5] = conj E[1], O[5] = conj O[1], W^5 = (-c1, s1)
    P[2] = er[2] + s1 * orr[2] - c1 * oi[2], Q[2] = ei[2] + s1 * oi[2] + c1 * orr[2];   // W^2 = (s1, c1)
    P[4] = er[2] - s1 * orr[2] + c1 * oi[2], Q[4] = -ei[2] + s1 * oi[2] + c1 * orr[2];  // W^4 = (-s1, c1)
}

template <int NC, bool DBG>
__global__ __launch_bounds__(kColsThreads, 2) void k_pfa_cols(ColsArgs A) {
    extern __shared__ __align__(16) unsigned char pfa_lds[];
    uint4 *s_coef = reinterpret_cast<uint4 *>(pfa_lds);
    float *s_bm = reinterpret_cast<float *>(pfa_lds + kCoefBytes);  // [output block][thread]: the lane's largest |y_d|^2 + |y_p|^2 of the block
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < kCoefFrags * 64; i += kColsThreads) s_coef[i] = A.coef[i];
    __syncthreads();
    const int ai = lane & 15, g = ai >> 2, r = ai & 3, ks = lane >> 4;
    const bool odd = lane & 1;
    auto t2_of = [&](int i) { return odd ? (i == 0 ? 6 : 12 - i) : i; };  // the output t2 behind slot i of this lane
    constexpr int kBlocks = kTiles;  // 196 blocks of 16 lags t3 per cell = the tiles of the inter-pass buffer
    const int qch = A.qchunk > 0 ? A.qchunk : 1, nq = (kBlocks + qch - 1) / qch;
    // work list: qch adjacent tiles of one cell (83 KB each, contiguous), then the same tiles of the NEXT cell: the workgroups that run
    // together work on different cells, so a cell's running maximum is settled by its first few waves (qch = 1: 2.4 % of the wave items go
    // through the values' pass, 4: 3.0 %, 8: 3.8 %)
    // (item = (q ncells + cl) qch + b walked in its three digits: the divisions of a 64-bit item number by run-time values were ~400 scalar
    //  instructions per item)
    const unsigned ncl = (unsigned)A.ncells, uq = (unsigned)qch;
    unsigned b = blockIdx.x % uq, cl = (blockIdx.x / uq) % ncl, q = blockIdx.x / (uq * ncl);
    const unsigned gb = gridDim.x % uq, gcl = (gridDim.x / uq) % ncl, gq = gridDim.x / (uq * ncl);
    for (; q < (unsigned)nq; b += gb, cl += gcl + (b >= uq ? (b -= uq, 1u) : 0u), q += gq + (cl >= ncl ? (cl -= ncl, 1u) : 0u)) {
        const int blk = (int)(q * uq + b);
        const int t0 = 16 * blk + 4 * wave;
        if (blk >= kBlocks || t0 >= K3) continue;
        const int cell = A.cell0 + cl;
        float *const lbp = A.lb + cell / A.lb_div;
        const int t3 = min(t0 + g, K3 - 1);
        const uint32_t *base = A.Bw + (size_t)cl * kCellElems;
        // ---- A fragments: [component][quad][ins], k1 = 4 mg .. 4 mg + 3 with mg = 4 ins + ks, of (k2 = 4 quad + r, t3)
        uint4 fa[2][3][4];
#pragma unroll
        for (int quad = 0; quad < 3; ++quad)
#pragma unroll
            for (int ins = 0; ins < 4; ++ins) {
                const int mg = 4 * ins + ks;
                uint4 l0 = make_uint4(0, 0, 0, 0), l1 = l0;
                const size_t off = bw_piece(0, 4 * quad + r, t3);
#ifdef PFA_EXP_C_NOLOAD
                l0 = make_uint4(0x3c003800u + lane, 0x38003c00u + cl + blk, 0x3c003400u + quad, 0x34003c00u + ins), l1 = l0;
                if (false) {
#else
                {
#endif
                // (rows past the 54th -- k1 >= 54: the K padding of the last matrix instruction -- read row pair 26 again: their coefficients are
                //  zeros and the buffer holds finite values, and a load under a lane condition costs a branch and a full vmcnt(0) wait per quad)
                l0 = *reinterpret_cast<const uint4 *>(base + (size_t)min(2 * mg, MP - 1) * K2 * (kTileLags * 4) + off);
                l1 = *reinterpret_cast<const uint4 *>(base + (size_t)min(2 * mg + 1, MP - 1) * K2 * (kTileLags * 4) + off);
                }
                fa[0][quad][ins] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                fa[1][quad][ins] = make_uint4(l0.z, l0.w, l1.z, l1.w);
            }
        // |y|^2 of output block nb: m2[c][i] for the lane's (t1, t3) -- each output in ONE lane of its (re, im) pair: the even lane holds
        // t2 = 0, 1, .., 5 (i = t2), the odd lane t2 = 6, 11, 10, .., 7 (i = 0, 1, .., 5: t2 = 12 - i)
        auto matrix = [&](int nb, f4 (&acc2)[NC][3], auto parts_tag) {
            constexpr int PARTS = decltype(parts_tag)::value;  // 2: coefficients hi + lo (the values), 1: hi only (the bound pass)
            uint4 fb[4][2];
#pragma unroll
            for (int ins = 0; ins < 4; ++ins)
#pragma unroll
                for (int part = 0; part < PARTS; ++part) fb[ins][part] = s_coef[((nb * 4 + ins) * 2 + part) * 64 + lane];
            // (six independent accumulator chains: both components)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int quad = 0; quad < 3; ++quad) acc2[c][quad] = (f4){0.f, 0.f, 0.f, 0.f};
#ifdef PFA_EXP_C_NOMFMA  // (timing experiments, tools/exp/r6_pfa_parts.sh: results INVALID)
#pragma unroll
                for (int quad = 0; quad < 3; ++quad) acc2[c][quad] = (f4){__uint_as_float(fa[c][quad][0].x), __uint_as_float(fb[1][0].y), __uint_as_float(fa[c][quad][2].z), __uint_as_float(fb[3][1].w)};
#else
#pragma unroll
                for (int ins = 0; ins < 4; ++ins)
#pragma unroll
                    for (int part = 0; part < PARTS; ++part)
#pragma unroll
                        for (int quad = 0; quad < 3; ++quad)
                            acc2[c][quad] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[c][quad][ins]), __builtin_bit_cast(h8, fb[ins][part]),
                                                                                   acc2[c][quad], 0, 0, 0);
#endif
            }
        };
        auto epilogue = [&](const f4 (&acc2)[NC][3], float (&m2)[2][6]) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f4(&acc)[3] = acc2[c];
#ifdef PFA_EXP_C_NOEPI
#pragma unroll
                for (int t = 0; t < 6; ++t) m2[c][t] = acc[t >> 1][t & 1] + acc[t >> 1][2 + (t & 1)];
                continue;
#endif
                // lane (G = lane >> 4, o = lane & 15): acc[quad][rr] = row 4 G + rr <-> (t3 = t0 + G, k2 = 4 quad + rr) of output 16 nb + o
                float v[12], P[7], Q[7];
#pragma unroll
                for (int k = 0; k < 12; ++k) v[k] = acc[k >> 2][k & 3];
                real_dft12(v, P, Q);
                // the lane pair (o even: the real part of z, o odd: the imaginary part) holds A = DFT12(re z) = P_e + j Q_e and B = DFT12(im z) =
                // P_o + j Q_o, and y[t] = A[t] + j B[t], y[12 - t] = conj(A[t]) + j conj(B[t]):
                //   |y[t]|^2 = S + X, |y[12 - t]|^2 = S - X with S = P_e^2 + Q_e^2 + P_o^2 + Q_o^2, X = 2 (Q_e P_o - P_e Q_o).
                // With x = Q P' - P Q' (' = the partner lane's) the even lane's S + 2 x is |y[t]|^2 and the odd lane's is |y[12 - t]|^2: six
                // instructions per t and pair of outputs, each output formed once (the first form squared re y and im y in both lanes).
                // (the compiler keeps a v_mov_b32_dpp in front of each of these; the 17 DPP-operand adds / products of a component issued by
                //  hand -- 34 instructions fewer per output block -- measured the same 0.96 ms: the kernel waits on latency, HISTORY.md 8)
                auto partner = [](float a) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xf, 0xf, true)); };
                {
                    const float p0 = P[0] * P[0], p6 = P[6] * P[6];  // Q[0] = Q[6] = 0
                    const float S0 = p0 + partner(p0), S6 = p6 + partner(p6);
                    m2[c][0] = odd ? S6 : S0;
                }
#pragma unroll
                for (int t = 1; t < 6; ++t) {
                    const float sl = fmaf(Q[t], Q[t], P[t] * P[t]);
                    const float S = sl + partner(sl);
                    float x = partner(P[t]) * Q[t];
                    x = fmaf(-partner(Q[t]), P[t], x);
                    m2[c][t] = fmaf(2.f, x, S);
                }
            }
        };
        auto block = [&](int nb, float (&m2)[2][6], auto parts_tag) {
            f4 acc2[NC][3];
            matrix(nb, acc2, parts_tag);
            epilogue(acc2, m2);
        };
        const int t3o = t0 + (lane >> 4);  // the lag t3 of this lane's outputs
        constexpr int kParts1 = DBG ? 2 : kBoundParts;  // (the debug instantiation reports this pass's values: both parts)
        float best = 0.f, ssum = 0.f;
        // The bound pass, software-pipelined: the matrix instructions of output block nb + 1 are in flight while the vector pipe works on
        // block nb's accumulators (two waves per SIMD that started together stay in step -- both in their matrix phase, then both in
        // their epilogue -- so the overlap has to come from inside the wave).  Fully unrolled: one scheduling region (0.93 -> 0.905 ms per
        // 201 cells; a forced 1 : 6 or 1 : 9 matrix : vector interleave by sched_group_barrier measured the same).
        f4 accp[2][NC][3];
        matrix(0, accp[0], std::integral_constant<int, kParts1>{});
        // the cell's maximum so far and the PRN's running bound; stale values are lower values: a redundant visit of the values' pass.
        // (Requested here, behind block 0's matrix instructions and their wait for the fragments: two device-scope loads in front of
        //  that wait would hold up the whole item, here they have six output blocks to arrive in.)
        const float lbv = __hip_atomic_load(lbp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the value half of the packed word alone: with the 64-bit load the compiler reuses the dead low register at once and waits for it)
        const unsigned cur = __hip_atomic_load(reinterpret_cast<const unsigned *>(A.cellmax + cell) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float m2[2][6];
            if (nb + 1 < NB) matrix(nb + 1, accp[(nb + 1) & 1], std::integral_constant<int, kParts1>{});
            epilogue(accp[nb & 1], m2);
            const int t1 = (16 * nb + (lane & 15)) >> 1;
            float bmax = 0.f, bsum = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float e = NC == 2 ? m2[0][i] + m2[1][i] : m2[0][i];
                bmax = fmaxf(bmax, e), bsum += e;
            }
            if (!(t1 < K1 && t3o < K3)) bmax = 0.f, bsum = 0.f;  // (selects, no branch: the seven blocks stay one scheduling region)
            best = fmaxf(best, bmax), ssum += bsum;
            s_bm[nb * kColsThreads + tid] = bmax;  // (read back by this lane only, in the rare values' pass)
            if (DBG && cell == A.dbg_cell && t0 / 4 == A.dbg_group && t1 < K1) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int i = 0; i < 6; ++i) A.dbg[((c * K1 + t1) * 12 + t2_of(i)) * 4 + (lane >> 4)] = m2[c][i];
            }
        }
        if (A.stats && lane == 0) atomicAdd(A.stats, 1ull);
        // Cauchy-Schwarz: (w_d |y_d| + w_p |y_p|)^2 <= (w_d^2 + w_p^2)(|y_d|^2 + |y_p|^2).  If even that bound, over all of the wave's
        // outputs, stays below both the cell's maximum so far and the sieve threshold of the PRN's running bound, the wave has nothing
        // to report (bds_acq_wcols.h).  Otherwise the exact values: the outputs are recomputed (they were never all in registers) --
        const float wsum2 = NC > 1 ? A.w0 * A.w0 + A.w1 * A.w1 : A.w0 * A.w0;
        float dlt = 0.f;  // what the values' pass may add to sqrt(|y_d|^2 + |y_p|^2) of this lane's outputs
        if (kParts1 == 1) {
            // the bound pass multiplies by fp16(coefficient) only (half the matrix work).  What the values' pass adds, per component:
            // |sum x lo| <= 2^-11 ||x||_1 (|lo| <= 2^-12 per real entry of the rotation), through the exact 12-point stage
            // <= 2^-11 sqrt(636) ||x||_2 over the 636 inputs of a lag t3, and ||x||_2 <= ||y_hi||_2 / (sqrt(636) - ||T_lo||) with
            // ||T_lo|| <= sqrt(12) * 53 * 2^-11.5 = 0.063: delta_c <= 4.9e-4 sqrt(sum over the lag's 636 outputs of |y_hi,c|^2), and by
            // Minkowski sqrt(|y_d|^2 + |y_p|^2) grows by at most sqrt(delta_d^2 + delta_p^2) = 4.9e-4 sqrt(sum of what `best` is the max of).
            float s16 = ssum;  // the 16 lanes of a lag: rotations within the DPP row (row_ror 8, 4, 2, 1), no LDS round trips
            s16 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s16), 0x128, 0xf, 0xf, true));
            s16 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s16), 0x124, 0xf, 0xf, true));
            s16 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s16), 0x122, 0xf, 0xf, true));
            s16 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s16), 0x121, 0xf, 0xf, true));
            dlt = 5.0e-4f * sqrtf(s16);
        }
        const float ub = wave_max_f32(sqrtf(best) + dlt);
        const float bw = ub * ub * wsum2 * 1.00001f;
        const float curv = __uint_as_float(cur), lim = fminf(curv, lbv * A.keep);
#if defined(PFA_EXP_C_NOEXACT) || defined(PFA_EXP_C_NOMFMA) || defined(PFA_EXP_C_NOEPI) || defined(PFA_EXP_C_NOLOAD) || defined(PFA_EXP_R_NOLOAD) || defined(PFA_EXP_R_NOSTORE)
        if (bw < 0.f)  // (timing experiments on invalid data: the bound pass alone)
#else
        if (!(bw < lim * lim))  // (wave-uniform; also taken while the bounds are unset or not finite)
#endif
        {
            if (A.stats && lane == 0) atomicAdd(A.stats + 1, 1ull);
            // -- only the output blocks in which some lane's bound reaches the limit (the same expression per lane and block as the wave's test
            // above: the block that failed it is among them); every output of the other blocks is below the cell's maximum so far and
            // below the list's threshold.  Typically one block of the seven.
            unsigned fmask = 0;
            for (int nb = 0; nb < NB; ++nb) {
                const float u = sqrtf(s_bm[nb * kColsThreads + tid]) + dlt;
                if (__builtin_amdgcn_ballot_w64(!(u * u * wsum2 * 1.00001f < lim * lim))) fmask |= 1u << nb;
            }
            if (A.stats && lane == 0) atomicAdd(A.stats + 3, (unsigned long long)__builtin_popcount(fmask));
            // -- a lane keeps the two largest of its values with their lags (first lag on ties, like max()).  Two qualifying values in one
            // lane's 42 outputs are the rare case of the rare case: then a third pass lists exhaustively.
            float top1 = -1.f, top2 = -1.f;
            int lag1 = 0x7fffffff, lag2 = 0x7fffffff;
            for (int nb = 0; nb < NB; ++nb) {
                if (!((fmask >> nb) & 1)) continue;
                float m2[2][6];
                block(nb, m2, std::integral_constant<int, 2>{});
                const int t1 = (16 * nb + (lane & 15)) >> 1;
                if (t1 < K1 && t3o < K3) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        // raw v_sqrt_f32 (1 ulp): the value only feeds the sieve; S + 2 x of a vanishing output may come out below zero
                        float a = A.w0 * __builtin_amdgcn_sqrtf(fmaxf(m2[0][i], 0.f));
                        if (NC > 1) a += A.w1 * __builtin_amdgcn_sqrtf(fmaxf(m2[NC - 1][i], 0.f));
                        const int lag = (int)lag_of(t1, t2_of(i), t3o);
                        if (a > top1 || (a == top1 && lag < lag1)) {
                            top2 = top1, lag2 = lag1, top1 = a, lag1 = lag;
                        } else if (a > top2 || (a == top2 && lag < lag2)) {
                            top2 = a, lag2 = lag;
                        }
                    }
                }
            }
            const float Mw = wave_max_f32(top1);
            if (Mw >= 0.f) {
                const float thr = fmaxf(Mw, lbv) * A.keep;
                const bool newmax = __float_as_uint(Mw) >= cur;  // this wave holds (a tie of) the cell's maximum so far
                const unsigned long long hit1 = __builtin_amdgcn_ballot_w64(top1 >= thr), hit2 = __builtin_amdgcn_ballot_w64(top2 >= thr);
                if (newmax || hit1) {
                    if (!hit2) {
                        const int total = __builtin_popcountll(hit1);
                        if (total > 0) {  // one reservation per wave on the list's counter
                            int base_i = 0;
                            if (lane == 0) base_i = atomicAdd(A.extra_count, total);
                            base_i = __builtin_amdgcn_readfirstlane(base_i);
                            if (top1 >= thr) {
                                const int idx = base_i + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hit1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hit1, 0u));
                                if ((unsigned)idx < (unsigned)A.extra_cap) {  // (unsigned: a counter run over 2^31 must not index backwards)
                                    Extra ex;
                                    ex.v = top1, ex.lag = lag1, ex.cell = cell;
                                    A.extra[idx] = ex;
                                }
                            }
                        }
                    } else {
                        if (A.stats && lane == 0) atomicAdd(A.stats + 2, 1ull);
                        for (int nb = 0; nb < NB; ++nb) {
                            if (!((fmask >> nb) & 1)) continue;
                            float m2[2][6];
                            block(nb, m2, std::integral_constant<int, 2>{});
                            const int t1 = (16 * nb + (lane & 15)) >> 1;
                            const bool mine = t1 < K1 && t3o < K3;
#pragma unroll
                            for (int i = 0; i < 6; ++i) {
                                float a = -1.f;
                                if (mine) {
                                    a = A.w0 * __builtin_amdgcn_sqrtf(fmaxf(m2[0][i], 0.f));
                                    if (NC > 1) a += A.w1 * __builtin_amdgcn_sqrtf(fmaxf(m2[NC - 1][i], 0.f));
                                }
                                const unsigned long long mask = __builtin_amdgcn_ballot_w64(a >= thr);
                                if (mask) {  // (wave-uniform)
                                    int base_i = 0;
                                    if (lane == 0) base_i = atomicAdd(A.extra_count, __builtin_popcountll(mask));
                                    base_i = __builtin_amdgcn_readfirstlane(base_i);
                                    if (a >= thr) {
                                        const int idx = base_i + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                        if ((unsigned)idx < (unsigned)A.extra_cap) {  // (unsigned: a counter run over 2^31 must not index backwards)
                                            Extra ex;
                                            ex.v = a, ex.lag = (int)lag_of(t1, t2_of(i), t3o), ex.cell = cell;
                                            A.extra[idx] = ex;
                                        }
                                    }
                                }
                            }
                        }
                    }
                    if (newmax) {
                        int bestlag = top1 == Mw ? lag1 : 0x7fffffff;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) bestlag = min(bestlag, __shfl_xor(bestlag, o));
                        if (lane == 0) {
                            atomicMax(A.cellmax + cell, wc_pack(Mw, bestlag));
                            if (Mw > lbv) atomicMax(reinterpret_cast<unsigned *>(lbp), __float_as_uint(Mw));
                        }
                    }
                }
            }
        }
    }
}

// ---- forward transforms: the spectra of the signal (one per call) and of the codes (cached), in the CRT layout ---------------------
// X[k1, k2, k3] = sum x[n] W_N^(-n k) with n = (n1 N/53 + n2 N/12 + n3 N/3125) mod N: rows over n3 (k_pfa_fwd_rows, the row pass's
// stages run on conjugates), then 53 points over n1 and 12 over n2 as plain sums (fp32; 1e8 complex products per transform).
template <class Loader>
__global__ __launch_bounds__(128) void k_pfa_fwd_rows(Loader ld, float2 *T /* [batch][636][3125] */) {
    __shared__ float2 region[K3 + 11];
    const int row = blockIdx.x, batch = blockIdx.y, j = threadIdx.x;
    const bool live = j < 125;
    const int jj = live ? j : 124;
    const long base = ((long)(row / K2) * (NP / K1) + (long)(row % K2) * (NP / K2)) % NP;
    v2f x[25];
#pragma unroll
    for (int q = 0; q < 25; ++q) {
        const float2 v = ld(batch, (base + (long)(jj + 125 * q) * (NP / K3)) % NP);
        x[q] = (v2f){v.x, -v.y};  // forward transform = conj(inverse transform of the conjugate)
    }
    pk_radix25(x);
    if (live) {
#pragma unroll
        for (int sl = 0; sl < 25; ++sl) {
            const int p = slot25_index(sl);
            region[25 * j + p] = to_f2(p ? pk_cmul(x[sl], unit((jj * p) % K3, K3)) : x[sl]);
        }
    }
    __syncthreads();
    const int si = jj % 25, spg = jj / 25;
    if (live) {
#pragma unroll
        for (int c5 = 0; c5 < 5; ++c5) {
            v2f z[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) z[r] = to_v2f(region[25 * (si + 25 * r) + 5 * spg + c5]);
            pk_radix5(z[0], z[1], z[2], z[3], z[4]);
#pragma unroll
            for (int u = 0; u < 5; ++u) region[25 * (si + 25 * u) + 5 * spg + c5] = to_f2(u ? pk_cmul(z[u], unit((si * u) % 125, 125)) : z[u]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 25; ++i) x[i] = to_v2f(region[25 * (i + 25 * (jj / 25)) + jj % 25]);
    pk_radix25(x);
    if (live) {
        float2 *o = T + ((size_t)batch * K1 * K2 + row) * K3;
#pragma unroll
        for (int sl = 0; sl < 25; ++sl) o[j + 125 * slot25_index(sl)] = make_float2(x[sl].x, -x[sl].y);
    }
}

// U[batch][k1][n2][k3] = sum_n1 T[batch][n1][n2][k3] W53^(-n1 k1)
__global__ __launch_bounds__(256) void k_pfa_fwd_53(const float2 *T, float2 *U) {
    __shared__ float2 w[K1];
    if (threadIdx.x < K1) {
        float sn, cs;
        sincospif(-2.0f * (float)threadIdx.x / (float)K1, &sn, &cs);
        w[threadIdx.x] = make_float2(cs, sn);
    }
    __syncthreads();
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (n2, k3)
    const int batch = blockIdx.y;
    if (e >= (long)K2 * K3) return;
    const float2 *t = T + (size_t)batch * NP + e;
    float2 *u = U + (size_t)batch * NP + e;
    float2 x[K1];
#pragma unroll
    for (int n1 = 0; n1 < K1; ++n1) x[n1] = t[(size_t)n1 * K2 * K3];
    for (int k1 = 0; k1 < K1; ++k1) {
        float ar = 0.f, ai = 0.f;
        int idx = 0;
#pragma unroll
        for (int n1 = 0; n1 < K1; ++n1) {
            const float2 ww = w[idx];
            ar = fmaf(x[n1].x, ww.x, fmaf(-x[n1].y, ww.y, ar));
            ai = fmaf(x[n1].x, ww.y, fmaf(x[n1].y, ww.x, ai));
            idx += k1;
            idx -= idx >= K1 ? K1 : 0;
        }
        u[(size_t)k1 * K2 * K3] = make_float2(ar, ai);
    }
}

// 12 points over n2, then the stored form: value * scale (conjugated for the code spectra) as fp16 complex;
// doubled = 1: signal spectrum, rows [k1][k2][2 x 3125]; 0: code spectra [batch][k1][k2][3125] from dst
__global__ __launch_bounds__(256) void k_pfa_fwd_12(const float2 *U, uint32_t *dst, long dst_batch_stride, int conj_flag, float scale, int doubled) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (k1, k3)
    const int batch = blockIdx.y;
    if (e >= (long)K1 * K3) return;
    const int k1 = (int)(e / K3), k3 = (int)(e % K3);
    const float2 *u = U + (size_t)batch * NP + (size_t)k1 * K2 * K3 + k3;
    float2 x[K2];
#pragma unroll
    for (int n2 = 0; n2 < K2; ++n2) x[n2] = u[(size_t)n2 * K3];
    constexpr float c[12] = {1.f, 0.86602540378443865f, 0.5f, 0.f, -0.5f, -0.86602540378443865f, -1.f, -0.86602540378443865f, -0.5f, 0.f, 0.5f, 0.86602540378443865f};
    constexpr float sn[12] = {0.f, 0.5f, 0.86602540378443865f, 1.f, 0.86602540378443865f, 0.5f, 0.f, -0.5f, -0.86602540378443865f, -1.f, -0.86602540378443865f, -0.5f};
    uint32_t *d = dst + (size_t)batch * dst_batch_stride;
#pragma unroll
    for (int k2 = 0; k2 < K2; ++k2) {
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int n2 = 0; n2 < K2; ++n2) {  // W12^(-n2 k2) = (c, -sn)[(n2 k2) mod 12]
            const float wr = c[(n2 * k2) % 12], wi = -sn[(n2 * k2) % 12];
            ar = fmaf(x[n2].x, wr, fmaf(-x[n2].y, wi, ar));
            ai = fmaf(x[n2].x, wi, fmaf(x[n2].y, wr, ai));
        }
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const v2f val = (v2f){ar * scale, (conj_flag ? -ai : ai) * scale};
        const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(val, h2));
        if (doubled) {
            d[((size_t)k1 * K2 + k2) * (2 * K3) + k3] = pk;
            d[((size_t)k1 * K2 + k2) * (2 * K3) + K3 + k3] = pk;
        } else {
            d[((size_t)k1 * K2 + k2) * K3 + k3] = pk;
        }
    }
}

// nb transforms: tmp holds 2 x nb x NP float2
template <class Loader>
inline void forward(hipStream_t st, Loader ld, int nb, float2 *tmp, uint32_t *dst, long dst_batch_stride, int conj_flag, float scale, int doubled) {
    float2 *T = tmp, *U = tmp + (size_t)nb * NP;
    hipLaunchKernelGGL(k_pfa_fwd_rows<Loader>, dim3(K1 * K2, nb), dim3(128), 0, st, ld, T);
    hipLaunchKernelGGL(k_pfa_fwd_53, dim3((K2 * K3 + 255) / 256, nb), dim3(256), 0, st, (const float2 *)T, U);
    hipLaunchKernelGGL(k_pfa_fwd_12, dim3((K1 * K3 + 255) / 256, nb), dim3(256), 0, st, (const float2 *)U, dst, dst_batch_stride, conj_flag, scale, doubled);
}

}  // namespace pfa
}  // namespace bds
