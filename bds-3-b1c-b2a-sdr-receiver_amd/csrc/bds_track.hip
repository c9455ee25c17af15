// Tracking: channel-batched correlate-and-dump + DLL/PLL loop update on the GPU.
//
// Replaces BDS-3_B2a/tracking.m:98-441, BDS-3_B1C/NB_tracking.m:107-448 and
// BDS-3_B1C/WB_tracking.m:114-488 (the per-channel, per-epoch MATLAB loops), plus the
// C/N0 + lock-detector post-pass of include/Calc_CNo_PLD.m.
//
// Structure: the window of the IF record the channels can touch lives in HBM (int8).  Every epoch is ONE launch:
//   k_trk_correlate  grid (workgroups, channels).  Head: every workgroup applies the loop update of the previous
//                    epoch to the channel state (apply_update: fixed-order sum of that epoch's partial sums, the
//                    discriminators and loop filters in f64 exactly in the reference's operation order; workgroup 0
//                    writes the result arrays and the new state; state and partial sums ping-pong between buffers).
//                    Body: the workgroup re-derives the epoch geometry (blksize, code / carrier NCO start values)
//                    from the f64 state and emits its 6/12/18 partial correlator sums -- correlate_runs (default:
//                    prefix sums of the carrier-wiped samples + an exact search of the code-index steps) or
//                    correlate_slice (per sample, BDS_TRK_PERSAMPLE)
//   k_trk_update     apply_update as its own launch: closes the last epoch (and every epoch with
//                    BDS_TRK_NOFUSE_UPDATE)
// The host enqueues all epochs back to back and never synchronises inside the loop: the
// sequential dependence (next blksize / phases depend on this epoch's discriminators) is
// carried entirely by device memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <type_traits>

#include "bds_debug.h"
#include "bds_internal.h"
#include "bds_strict_math.h"

namespace bds {

// samples per correlate workgroup (TrkParams::chunk): 8192 for the 10-ms B1C epochs, 2048 for the
// 1-ms B2a epochs (measured on 12 channels at 99.375 MS/s, us per epoch: B1C WB 148 / 93 / 84 / 82 and
// B2a 23.3 / 20.2 / 21.7 / 29.7 at 1024 / 2048 / 4096 / 8192)
static constexpr int kTrkThreads = 256;
static constexpr size_t kDataSlack = 256;  // bytes past the record a segment's aligned dword reads may touch
static constexpr int kNSums = 18;      // I_E,Q_E,I_P,Q_P,I_L,Q_L x {data, pilot BOC11 / B2a pilot, pilot BOC61}

struct ChanState {
    double codeFreq, remCodePhase, carrFreq, carrFreqBasis, remCarrPhase;
    double oldCodeNco, oldCodeError, d2CarrError, dCarrError;
    double codeFreqBasis;  // channel.codeFreq (tracking.m:389)
    long long pos;         // sample offset of the next read: ftell / dataAdaptCoeff (tracking.m:226)
    int prn;               // 0 = channel unused
    int active;            // 1 while the channel keeps tracking; 0 stopped at a short read (end of file);
                           // -2 its reads left the loaded window of the record (the host reloads the whole file)
    int completed;         // epochs finished
    int pad;
};

static constexpr int kUpdGroups = 14;    // groups of correlate workgroups in the fixed-order sum of the partial sums
static constexpr int kUpdThreads = 256;  // >= kUpdGroups * kNSums
static constexpr int kUpdScratch = (1 + kUpdGroups) * kNSums * 8 + 48 + (int)sizeof(ChanState);  // LDS bytes of apply_update

struct TrkParams {
    int mode;        // BDS_TRACK_*
    int pilot;       // pilot correlators on
    int cplx;        // fileType 2: the record is interleaved I/Q int8 pairs (tracking.m:132-136,242-246)
    int chunk;       // samples per correlate workgroup
    int runs;        // 1: run-based correlator (correlate_runs), chunk = kTrkThreads * 8 or * 16
    int prec;        // numerics of the run-based correlator's carrier wipe-off and prefix sums (Tuning::trk_prec)
    int code_len;    // 10230
    int n_epochs;
    double fs, inv_fs;
    double spacing;  // dllCorrelatorSpacing (earlyLateSpc)
    double tau1, tau2, pdi, pf1, pf2, pf3, factor;
    long long n_bytes;  // samples in the record: file bytes / dataAdaptCoeff
    long long base;     // first sample of the record held in HBM (only the window the channels can touch is loaded)
    long long win_end;  // one past the last sample held
};

struct TrkOut {  // device arrays [n_ch][n_epochs]
    double *absoluteSample, *codeFreq, *carrFreq, *I_P, *I_E, *I_L, *Q_E, *Q_P, *Q_L;
    double *Pilot_I_P, *Pilot_Q_P, *Pilot_I_E, *Pilot_I_L, *Pilot_Q_E, *Pilot_Q_L;
    double *dllDiscr, *dllDiscrFilt, *pllDiscr, *pllDiscrFilt, *remCodePhase, *remCarrPhase;
};

// Padded-array look-up of the reference: [c(L) c(1..L) c(1)], 1-based index i in 1..L+2
// (B2a/tracking.m:158; B1C/WB_tracking.m:181,187,192).  The device keeps, per PRN, the three arrays
// the trackers index -- data code, pilot code, pilot BOC(6,1) -- expanded to their own resolution
//   B2a: chips;  B1C: BOC(1,1) half-chips [-c, +c] (generateDataBOC11.m:85-91);
//   BOC(6,1): twelfths (-1)^ii c, ii = 1..12 (generatePilotBOC61.m:89-96)
// with kTabPad wrapped entries on both sides, so a look-up is one clamped index and one byte load
// (deriving chip, sub-chip sign and wrap from the index cost ~11 instructions per look-up, nine
// look-ups per sample in wide-band mode).
static constexpr int kTabPad = 32;
static constexpr long kTabStride = 122880;  // >= 12 * 10230 + 2 * kTabPad, multiple of 64
__device__ __forceinline__ float tab_at(const int8_t *__restrict__ tab, int n_units, int i1) {
    int j = i1 + (kTabPad - 2);  // unit index i1 - 2, shifted by the left padding
    BDS_DASSERT(j >= 0 && j <= n_units + 2 * kTabPad - 1);  // (the clamp below is a guard, never the intended look-up)
    j = max(0, min(j, n_units + 2 * kTabPad - 1));
    return (float)tab[j];
}
// data and pilot codes are read at the same index: slot 0 of a PRN holds them interleaved
__device__ __forceinline__ char2 tab2_at(const int8_t *__restrict__ tab, int n_units, int i1) {
    int j = i1 + (kTabPad - 2);
    BDS_DASSERT(j >= 0 && j <= n_units + 2 * kTabPad - 1);
    j = max(0, min(j, n_units + 2 * kTabPad - 1));
    return reinterpret_cast<const char2 *>(tab)[j];
}

// The reference's replica index vectors are MATLAB colon vectors,
//     tcode = (rem -+ spc)[*2] : step[*2] : ((blksize-1)*step + rem -+ spc)[*2]     (tracking.m:260-286, NB_tracking.m:271-297,
//                                                                                      WB_tracking.m:289-317),
// and MATLAB does not build a:d:b as a + k d throughout (MathWorks' published colonop.m, Technical Solution 1-4FLI96): with
// n = round((b-a)/d) intervals (one less if a + n d overshoots b by more than tol = 2 eps max(|a|,|b|)) and the right end
// c = a + n d snapped to b within tol, elements 0 .. floor(n/2) are a + k d, the elements after them c - (n-k) d -- generated from
// the RIGHT end --, and the mid-point of an even n is (a+c)/2.  The second half differs from a + k d by 0-2 ulp of tcode, which
// moves ceil() for a sample that sits on a chip boundary to rounding (round 6; oracle/matlab.py m_colon is the checker's copy).
struct ColonVec {
    double a, d, c;
    int n, h;
    bool even;
};
__device__ __forceinline__ ColonVec colon_vec(double a, double d, double b) {
    ColonVec v;
    const double tol = 2.0 * 2.220446049250313e-16 * fmax(fabs(a), fabs(b));
    long n = (long)floor((b - a) / d + 0.5);  // MATLAB round() of a non-negative quotient (the tcode vectors ascend)
    if (a + (double)n * d - b > tol) n -= 1;
    double c = a + (double)n * d;
    if (c - b > -tol) c = b;
    v.a = a, v.d = d, v.c = c, v.n = (int)n, v.h = (int)(n / 2), v.even = (n & 1) == 0;
    return v;
}
__device__ __forceinline__ double colon_at(const ColonVec &v, int k) {
    const double fwd = v.a + (double)k * v.d, bwd = v.c - (double)(v.n - k) * v.d;
    return k > v.h ? bwd : (v.even && k == v.h) ? (v.a + v.c) / 2 : fwd;
}
// the three replica vectors of an epoch: E, P, L
template <int SCALE2>
__device__ __forceinline__ void epoch_colons(double rem, double step, double spacing, long blk, ColonVec cv[3]) {
    const double sc = SCALE2 ? 2.0 : 1.0;
    const double last = (double)(blk - 1) * step + rem;  // (blksize-1)*codePhaseStep + remCodePhase, then -+ earlyLateSpc, then *2
    cv[0] = colon_vec((rem - spacing) * sc, step * sc, (last - spacing) * sc);
    cv[1] = colon_vec(rem * sc, step * sc, last * sc);
    cv[2] = colon_vec((rem + spacing) * sc, step * sc, (last + spacing) * sc);
    BDS_DASSERT(cv[0].n == blk - 1 && cv[1].n == blk - 1 && cv[2].n == blk - 1);  // MATLAB would stop at tcode(blksize) otherwise
}

struct EpochGeom {
    long long pos;
    long blk;
    double step, rem, carrFreq, remCarr;
};

__device__ __forceinline__ EpochGeom epoch_geom(const ChanState &s, const TrkParams &p) {
    EpochGeom g;
    g.pos = s.pos;
    g.step = s.codeFreq / p.fs;                                               // tracking.m:230
    g.blk = (long)ceil(((double)p.code_len - s.remCodePhase) / g.step);       // :233
    g.rem = s.remCodePhase;
    g.carrFreq = s.carrFreq;
    g.remCarr = s.remCarrPhase;
    return g;
}

// One block of samples [k0, k1) of one channel: 18 partial sums (f64) to part[].
template <int MODE>
__device__ __forceinline__ void correlate_slice(const int8_t *__restrict__ data, const int8_t *__restrict__ prim_d,
                                                const int8_t *__restrict__ prim_p, const TrkParams &p,
                                                const EpochGeom &g, long k0_first, long k_stride, bool pilot, double *sums) {
    constexpr int UNITS = MODE == BDS_TRACK_B2A ? 1 : 2;
    const int NU = UNITS * p.code_len;
    ColonVec cv[3];  // tracking.m:260-286 / WB_tracking.m:289-317
    epoch_colons<(MODE != BDS_TRACK_B2A)>(g.rem, g.step, p.spacing, g.blk, cv);
    const double two_pi = 6.283185307179586476925286766559;
    const double cyc0 = g.remCarr / two_pi;
    float acc[kNSums];
#pragma unroll
    for (int i = 0; i < kNSums; ++i) acc[i] = 0.f;
    double wr, wi, cr = 1.0, ci = 0.0;
    {
        const double dcyc = g.carrFreq * ((double)blockDim.x * p.inv_fs);
        sincospi(2.0 * (dcyc - floor(dcyc)), &wi, &wr);
    }
    int it = 0;
    const int8_t *__restrict__ dwin = data - (p.base * (p.cplx ? 2 : 1));  // data[] starts at sample p.base of the record
    // chunk k0 .. k0+chunk of the block, then (only when blksize outgrew the grid the call was sized for:
    // a code rate more than 2 % below the slowest channel's initial one) every k_stride-th chunk after it
    for (long k0 = k0_first; k0 < g.blk; k0 += k_stride) {
    const long k1 = min(g.blk, k0 + p.chunk);
    for (int k = (int)k0 + (int)threadIdx.x; k < (int)k1; k += (int)blockDim.x) {  // blksize < 2^31
        float raw, raw_q = 0.f;
        BDS_DASSERT(g.pos + k >= p.base && g.pos + k < p.win_end);  // inside the window of the record held in HBM
        if (p.cplx) {  // rawSignal = data(1:2:end) + 1i*data(2:2:end)  (tracking.m:242-246)
            const char2 v = reinterpret_cast<const char2 *>(dwin)[g.pos + k];
            raw = (float)v.x;
            raw_q = (float)v.y;
        } else {
            raw = (float)dwin[g.pos + k];
        }
        const double kd = (double)k;
        const double te = colon_at(cv[0], k), tp = colon_at(cv[1], k), tl = colon_at(cv[2], k);
        const int ie = (int)ceil(te) + 1, il = (int)ceil(tl) + 1, ip = (int)ceil(tp) + 1;
        // carrier: trigarg = (carrFreq*2*pi)*(k/fs) + remCarrPhase  (tracking.m:303-304).  A thread's
        // samples are blockDim apart, so its carrier is an f64 phasor rotated by a constant angle; it is
        // re-evaluated exactly (phase reduced in f64, cycles) every 8th step.
        if ((it & 7) == 0) {
            const double cyc = g.carrFreq * (kd * p.inv_fs) + cyc0;
            sincospi(2.0 * (cyc - floor(cyc)), &ci, &cr);
        } else {
            const double nr = cr * wr - ci * wi;
            ci = cr * wi + ci * wr;
            cr = nr;
        }
        ++it;
        const float c2 = (float)cr, s2 = (float)ci;
        float ib, qb;
        if (MODE == BDS_TRACK_B2A) {  // exp(+j th): q = real, i = imag (tracking.m:309-314)
            qb = raw * c2 - raw_q * s2;
            ib = raw * s2 + raw_q * c2;
        } else {  // exp(-j th): i = real, q = imag (NB_tracking.m:320-325)
            ib = raw * c2 + raw_q * s2;
            qb = raw_q * c2 - raw * s2;
        }
        const char2 ve = tab2_at(prim_d, NU, ie), vp = tab2_at(prim_d, NU, ip), vl = tab2_at(prim_d, NU, il);
        const float ce = (float)ve.x, cp = (float)vp.x, cl = (float)vl.x;
        acc[0] = fmaf(ce, ib, acc[0]);
        acc[1] = fmaf(ce, qb, acc[1]);
        acc[2] = fmaf(cp, ib, acc[2]);
        acc[3] = fmaf(cp, qb, acc[3]);
        acc[4] = fmaf(cl, ib, acc[4]);
        acc[5] = fmaf(cl, qb, acc[5]);
        if (pilot) {
            const float pe = (float)ve.y, pp = (float)vp.y, pl = (float)vl.y;
            acc[6] = fmaf(pe, ib, acc[6]);
            acc[7] = fmaf(pe, qb, acc[7]);
            acc[8] = fmaf(pp, ib, acc[8]);
            acc[9] = fmaf(pp, qb, acc[9]);
            acc[10] = fmaf(pl, ib, acc[10]);
            acc[11] = fmaf(pl, qb, acc[11]);
            if (MODE == BDS_TRACK_WB) {  // pilotBOC61(ceil(tcode*6)+1)  (WB_tracking.m:298,311,324)
                const int8_t *p6 = prim_p;  // the PRN's BOC(6,1) array
                const float se = tab_at(p6, 12 * p.code_len, (int)ceil(te * 6) + 1);
                const float sp = tab_at(p6, 12 * p.code_len, (int)ceil(tp * 6) + 1);
                const float sl = tab_at(p6, 12 * p.code_len, (int)ceil(tl * 6) + 1);
                acc[12] = fmaf(se, ib, acc[12]);
                acc[13] = fmaf(se, qb, acc[13]);
                acc[14] = fmaf(sp, ib, acc[14]);
                acc[15] = fmaf(sp, qb, acc[15]);
                acc[16] = fmaf(sl, ib, acc[16]);
                acc[17] = fmaf(sl, qb, acc[17]);
            }
        }
    }
    }
    // wave reduction (f64 from here), then one LDS hop
    __shared__ double s_part[kTrkThreads / 64][kNSums];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kNSums; ++i) {
        double v = (double)acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) s_part[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double v = 0;
        for (int w = 0; w < kTrkThreads / 64; ++w) v += s_part[w][threadIdx.x];
        sums[threadIdx.x] = v;
    }
}


// ---- run-based correlator ---------------------------------------------------------------------
// The replica codes are piecewise constant: at 99.375 MS/s a BOC(1,1) half-chip lasts 48.6 samples,
// a BOC(6,1) twelfth 8.1, a B2a chip 9.7.  With S(k) = the running sum of the carrier-wiped samples,
//     sum_k x[k] c[idx(k)]  =  c[idx(last)] S(end)  -  sum over index steps u (c[u] - c[u-1]) S(k_u),
// k_u = the first sample whose index reaches u.  The per-sample index of the reference, ceil() of element k of its colon
// vector (ColonVec above; x 6 for BOC(6,1), WB_tracking.m:298), is monotone in k -- neighbouring elements are one step apart to
// a few ulp in both halves and across the junction --, so k_u is found
// from a division and confirmed with the reference's own expression on k_u - 1 and k_u: every sample gets
// exactly the index the per-sample evaluation gives it, at two exact evaluations per code unit instead
// of one per sample and replica.
//   phase 1  thread t wipes the carrier off SEG consecutive samples (per-thread f64 phasor x a table
//            of the SEG per-sample rotations), writes the exclusive fp32 prefix sums of its segment to
//            LDS; the segment totals are scanned in f64 over the workgroup (segment bases)
//   phase 2  the index steps of the six (E/P/L x {code, BOC(6,1)}) sequences are spread over the
//            threads; each finds its k_u, reads S(k_u) = base + local prefix and adds its term in f64
#ifndef BDS_TRK_CAP1
#define BDS_TRK_CAP1 256
#endif
#ifndef BDS_TRK_CAP6
#define BDS_TRK_CAP6 1280
#endif
#ifndef BDS_TRK_MINW
#define BDS_TRK_MINW 1
#endif
static constexpr int kCap1 = BDS_TRK_CAP1, kCap6 = BDS_TRK_CAP6;  // code / BOC(6,1) table entries of a wave's pass staged in LDS
template <int R6>
__device__ __forceinline__ double code_arg(const ColonVec &v, int k) {
    double t = colon_at(v, k);  // element k of the reference's colon vector (-ffp-contract=off: one rounding per operation)
    if (R6) t = t * 6;
    return t;
}

// first k in (k_lo, k_hi] with ceil(arg(k)) >= u, given ceil(arg(k_lo)) < u <= ceil(arg(k_hi))
template <int R6>
__device__ __forceinline__ int first_sample_of(const ColonVec &v, double inv_inc, int u, int k_lo, int k_hi) {
    const double thr = (double)(u - 1);  // ceil(t) >= u  <=>  t > u - 1
    const double tgt = R6 ? thr * (1.0 / 6.0) : thr;  // a prediction only: confirmed below with the exact expression
    int kc = (int)floor((tgt - v.a) * inv_inc) + 1;
    kc = max(k_lo + 1, min(kc, k_hi));
    while (kc > k_lo + 1 && code_arg<R6>(v, kc - 1) > thr) --kc;
    while (kc < k_hi && !(code_arg<R6>(v, kc) > thr)) ++kc;
    return kc;
}

// LDS of one wave of correlate_runs<., SEG>: prefix sums, segment bases, rotation table, code-table slices
// (prec: TrkParams::prec -- 0 fp32 carrier + fp32 local prefix sums, 1 f64 prefix sums, 2 f64 carrier too, 3 the
// reference's own trigarg per sample; f64 prefix entries are 16 bytes, the f64 rotation table as well)
__host__ __device__ constexpr size_t runs_wave_lds(int seg, int prec) {
    return (size_t)(64 * seg + 64) * (prec >= 1 ? 16 : 8) + 64 * 16 + (size_t)seg * (prec >= 2 ? 16 : 8) + (size_t)kCap1 * 2 + kCap6;
}
static inline size_t runs_lds_bytes(int seg, int prec) {
    return std::max<size_t>(runs_wave_lds(seg, prec) * (kTrkThreads / 64), sizeof(double) * (kTrkThreads / 64) * kNSums);
}
__device__ __forceinline__ void wave_sync() {  // LDS written by this wave is read by other lanes of this wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 64-lane inclusive f64 add-scan on the DPP network (row shifts inside the 16-lane rows, then the row
// broadcasts): 3 instructions per step instead of two LDS-pipe permutes; lane 63 ends up with the total
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, BOUND);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, BOUND);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan_incl(double v) {
    v += dpp_f64<0x111, 0xf, true>(v);   // row_shr:1
    v += dpp_f64<0x112, 0xf, true>(v);   // row_shr:2
    v += dpp_f64<0x114, 0xf, true>(v);   // row_shr:4
    v += dpp_f64<0x118, 0xf, true>(v);   // row_shr:8
    v += dpp_f64<0x142, 0xa, false>(v);  // row_bcast:15 into rows 1, 3
    v += dpp_f64<0x143, 0xc, false>(v);  // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ double lane_f64(double v, int l) {  // broadcast of lane l (compile-time constant)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// The waves of a workgroup work independently (wave w on samples k0 + w 64 SEG .. of each chunk, its own LDS
// slice, no workgroup barrier before the final reduction): the kernel is bound by latency -- HBM reads,
// dependent f64 evaluations -- and independent waves hide it where barrier-separated phases cannot.
template <int MODE, int SEG, bool CPLX, int PREC>
__device__ __forceinline__ void correlate_runs(const int8_t *__restrict__ data, const int8_t *__restrict__ prim_d,
                                               const int8_t *__restrict__ prim_p, const TrkParams &p,
                                               const EpochGeom &g, long k0_first, long k_stride, bool pilot, double *sums) {
    constexpr int NT = kTrkThreads, NW = NT / 64, WCH = 64 * SEG;
    constexpr double scale = MODE == BDS_TRACK_B2A ? 1.0 : 2.0;
    constexpr int UNITS = MODE == BDS_TRACK_B2A ? 1 : 2;
    constexpr int NWD = SEG / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char trk_lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // PT: type of a segment's local prefix sums; CT: type of the carrier replica and the wiped samples
    using PT = std::conditional_t<(PREC >= 1), double, float>;
    using PT2 = std::conditional_t<(PREC >= 1), double2, float2>;
    using CT = std::conditional_t<(PREC >= 2), double, float>;
    using CT2 = std::conditional_t<(PREC >= 2), double2, float2>;
    constexpr size_t kLoc = (size_t)(WCH + 64) * sizeof(PT2);
    unsigned char *wl = trk_lds + runs_wave_lds(SEG, PREC) * wave;
    PT2 *s_loc = reinterpret_cast<PT2 *>(wl);                                       // [WCH + 64]: position i at i + i / SEG
    double2 *s_base = reinterpret_cast<double2 *>(wl + kLoc);                       // [64] segment bases
    CT2 *s_w = reinterpret_cast<CT2 *>(wl + kLoc + 64 * 16);                        // [SEG]
    char2 *s_t1 = reinterpret_cast<char2 *>(wl + kLoc + 64 * 16 + (size_t)SEG * sizeof(CT2));  // [kCap1]
    int8_t *s_t6 = reinterpret_cast<int8_t *>(s_t1 + kCap1);                                    // [kCap6]
    const int NU = UNITS * p.code_len, n6 = 12 * p.code_len;
    const double inc = g.step * scale, inv_inc = 1.0 / inc;
    ColonVec cv3[3];  // E, P, L
    epoch_colons<(MODE != BDS_TRACK_B2A)>(g.rem, g.step, p.spacing, g.blk, cv3);
    const double two_pi = 6.283185307179586476925286766559;
    const double cyc0 = g.remCarr / two_pi;
    constexpr int coeff = CPLX ? 2 : 1;
    constexpr int nwd = CPLX ? 2 * NWD : NWD;  // dwords of a segment
    const int8_t *__restrict__ dwin = data - p.base * coeff;
    double acc[kNSums];
#pragma unroll
    for (int i = 0; i < kNSums; ++i) acc[i] = 0.0;
    if ((PREC < 3 || PREC == 5) && lane < SEG) {  // rotation of sample j of a segment against its first sample
        const double cyc = g.carrFreq * ((double)lane * p.inv_fs);
        double sn, cs;
        sincospi(2.0 * (cyc - floor(cyc)), &sn, &cs);
        s_w[lane].x = (CT)cs, s_w[lane].y = (CT)sn;
    }
    // PREC 3: the carrier argument of every sample exactly as the reference forms it,
    //   trigarg = (carrFreq*2*pi) .* ((0:blksize) ./ fs) + remCarrPhase   (tracking.m:303-304; left to right, one rounding
    // per operation: the translation unit is built with -ffp-contract=off), then sin / cos of that f64 value
    const double w_ref = (g.carrFreq * 2.0) * 3.14159265358979323846;
    // the segment's bytes (I/Q pairs: 2 SEG bytes), whole aligned dwords around it
    uint32_t raw[nwd + 1];
    auto fetch = [&](long kw) {
        const long kb = kw + (long)lane * SEG;
        if (kb < g.blk) {
            BDS_DASSERT(g.pos + kb >= p.base && g.pos + kb < p.win_end);  // first sample of the segment inside the window held in HBM
            const uintptr_t a = (uintptr_t)(dwin + (g.pos + kb) * coeff);
            const uint32_t *__restrict__ q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
#pragma unroll
            for (int i = 0; i <= nwd; ++i) raw[i] = q[i];
        }
    };
    const long kw_first = k0_first + (long)wave * WCH;
    if (kw_first < g.blk) fetch(kw_first);
    // carrier at this lane's first sample of the first pass (tracking.m:303-304); every further pass is k_stride
    // samples on: one f64 rotation instead of a sincospi per pass
    double bs, bc, rs = 0.0, rc = 1.0;
    {
        const double cyc = g.carrFreq * ((double)(kw_first + (long)lane * SEG) * p.inv_fs) + cyc0;
        sincospi(2.0 * (cyc - floor(cyc)), &bs, &bc);
        if (kw_first + k_stride < g.blk) {  // (wave-uniform) a second pass exists
            const double cyr = g.carrFreq * ((double)k_stride * p.inv_fs);
            sincospi(2.0 * (cyr - floor(cyr)), &rs, &rc);
        }
    }
    for (long kwl = kw_first; kwl < g.blk; kwl += k_stride) {
        const int k0 = (int)kwl, k1 = (int)min(g.blk, kwl + WCH);  // blksize < 2^31
        wave_sync();  // s_w written / the previous pass's phase 2 done with the LDS slice
        // ---- index ranges of this pass's E/P/L replicas and their slice of the code tables (phase 2 then reads LDS
        // instead of issuing dependent global loads; longer ranges -- low sampling rates -- stay in global memory)
        int ua1[3], ub1[3], ua6[3], ub6[3];
        {  // the twelve range ends are wave-uniform: lane 3 e + ph (+ 6 for BOC(6,1)) evaluates one, readlane spreads them
            const int ph = lane % 3, end = (lane / 3) & 1, r6 = (lane / 6) & 1;
            const ColonVec &cl = ph == 0 ? cv3[0] : ph == 1 ? cv3[1] : cv3[2];
            double v = colon_at(cl, end ? k1 - 1 : k0);  // as code_arg
            if (r6) v = v * 6;
            const int idx = (int)ceil(v);
#pragma unroll
            for (int ph2 = 0; ph2 < 3; ++ph2) {
                ua1[ph2] = __builtin_amdgcn_readlane(idx, ph2), ub1[ph2] = __builtin_amdgcn_readlane(idx, 3 + ph2);
                ua6[ph2] = __builtin_amdgcn_readlane(idx, 6 + ph2), ub6[ph2] = __builtin_amdgcn_readlane(idx, 9 + ph2);
            }
        }
        const int lo1 = min(ua1[0], min(ua1[1], ua1[2])), hi1 = max(ub1[0], max(ub1[1], ub1[2])) + 1;
        const int lo6 = min(ua6[0], min(ua6[1], ua6[2])), hi6 = max(ub6[0], max(ub6[1], ub6[2])) + 1;
        const bool use1 = hi1 - lo1 < kCap1, use6 = MODE == BDS_TRACK_WB && pilot && hi6 - lo6 < kCap6;
        if (use1)
            for (int i = lane; i <= hi1 - lo1; i += 64) s_t1[i] = tab2_at(prim_d, NU, lo1 + i);
        if (use6)
            for (int i = lane; i <= hi6 - lo6; i += 64) s_t6[i] = (int8_t)tab_at(prim_p, n6, lo6 + i);
        // ---- phase 1: wipe the carrier off this lane's SEG samples, exclusive prefix sums of the segment to LDS
        const int kb = k0 + lane * SEG;
        const int n_here = max(0, min(SEG, k1 - kb));
        PT run_i = 0, run_q = 0;
        if (n_here > 0) {
            uint32_t wr[nwd];
            {
                const uint32_t sh = (uint32_t)((uintptr_t)(dwin + (g.pos + kb) * coeff) & 3);
#pragma unroll
                for (int i = 0; i < nwd; ++i) wr[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], sh);
            }
            const CT bcf = (CT)bc, bsf = (CT)bs;
            const int lbase = lane * SEG + lane;
            double trig5 = 0.0, c5 = 1.0, s5 = 0.0;
            const double dlt5 = w_ref * p.inv_fs;
            if constexpr (PREC == 5) {  // the reference's trigarg of the segment's first sample and its phasor
                trig5 = (w_ref * div_by_fs((double)kb, p.fs, p.inv_fs)) + g.remCarr;
                sincos_strict(trig5, s5, c5);
            }
            // sample j of the segment, carrier-wiped: (ib, qb).  PREC < 3: the carrier is base x table entry with
            // explicit FMAs (the exact two-rounding rule of the build only matters for the code index), in fp32
            // (PREC 0, 1) or f64 (PREC 2); PREC 3: sin / cos of the reference's own trigarg(k)
            auto wiped = [&](int j, CT &ib, CT &qb) {
                CT rw, rw_q = 0;
                if (CPLX) {  // rawSignal = data(1:2:end) + 1i*data(2:2:end)  (tracking.m:242-246)
                    const uint32_t w = wr[j >> 1];
                    rw = (CT)(int)(int8_t)(w >> ((j & 1) * 16));
                    rw_q = (CT)(int)(int8_t)(w >> ((j & 1) * 16 + 8));
                } else {
                    rw = (CT)(int)(int8_t)(wr[j >> 2] >> ((j & 3) * 8));
                }
                CT c2, s2;
                if constexpr (PREC == 3) {
                    const double tt = (double)(kb + j) / p.fs;
                    const double trig = (w_ref * tt) + g.remCarr;
                    sincos(trig, &s2, &c2);
                } else if constexpr (PREC == 5) {
                    // the same value from one library-free sin / cos per lane and pass: trigarg(k + j) = trigarg(k) + D_j with D_j
                    // EXACT (difference of two neighbouring f64 values), D_j = j dlt + eps_j with dlt = 2 pi carrFreq / fs and
                    // |eps_j| ~ 1e-10 rad the reference's own rounding noise, so
                    //   exp(i trigarg(k + j)) = exp(i trigarg(k)) x exp(i j dlt) x (1 + i eps_j)      (eps^2 / 2 < 1e-19)
                    // -- the segment's first phasor, the wave's rotation table, a first-order correction: 14 f64 operations per
                    // sample instead of 40
                    const double tt = div_by_fs((double)(kb + j), p.fs, p.inv_fs);
                    const double trig = (w_ref * tt) + g.remCarr;
                    const double eps = fma(-(double)j, dlt5, trig - trig5);
                    const CT2 w = s_w[j];
                    const double c1 = fma(c5, w.x, -(s5 * w.y)), s1 = fma(s5, w.x, c5 * w.y);
                    c2 = fma(-eps, s1, c1), s2 = fma(eps, c1, s1);
                } else if constexpr (PREC == 4) {
                    const double tt = div_by_fs((double)(kb + j), p.fs, p.inv_fs);
                    BDS_DASSERT(tt == (double)(kb + j) / p.fs);
                    BDS_DASSERT(tt == (double)(kb + j) / p.fs);
                    const double trig = (w_ref * tt) + g.remCarr;
                    sincos_strict(trig, s2, c2);
                } else {
                    const CT2 w = s_w[j];
                    c2 = fma(bcf, w.x, -(bsf * w.y)), s2 = fma(bsf, w.x, bcf * w.y);
                }
                if (MODE == BDS_TRACK_B2A) {  // exp(+j th): q = real, i = imag (tracking.m:309-314)
                    qb = CPLX ? fma(rw, c2, -(rw_q * s2)) : rw * c2;
                    ib = CPLX ? fma(rw, s2, rw_q * c2) : rw * s2;
                } else {  // exp(-j th): i = real, q = imag (NB_tracking.m:320-325)
                    ib = CPLX ? fma(rw, c2, rw_q * s2) : rw * c2;
                    qb = CPLX ? fma(rw_q, c2, -(rw * s2)) : -(rw * s2);
                }
            };
            if (n_here == SEG) {  // every segment but the last one of the block
#pragma unroll
                for (int j = 0; j < SEG; ++j) {
                    CT ib, qb;
                    wiped(j, ib, qb);
                    s_loc[lbase + j].x = run_i, s_loc[lbase + j].y = run_q;
                    run_i += (PT)ib, run_q += (PT)qb;
                }
            } else {
#pragma unroll
                for (int j = 0; j < SEG; ++j) {
                    CT ib, qb;
                    wiped(j, ib, qb);
                    s_loc[lbase + j].x = run_i, s_loc[lbase + j].y = run_q;
                    run_i += j < n_here ? (PT)ib : (PT)0, run_q += j < n_here ? (PT)qb : (PT)0;
                }
            }
        }
        if (kwl + k_stride < g.blk) {
            fetch(kwl + k_stride);  // next pass's bytes: in flight during the scan and phase 2
            const double nc = bc * rc - bs * rs;
            bs = bs * rc + bc * rs;
            bc = nc;
        }
        // exclusive f64 scan of the segment totals over the wave
        const double vi = wave_scan_incl((double)run_i), vq = wave_scan_incl((double)run_q);
        s_base[lane] = make_double2(vi - (double)run_i, vq - (double)run_q);
        const double ti = lane_f64(vi, 63), tq = lane_f64(vq, 63);  // pass totals
        wave_sync();
        // ---- phase 2
        auto prefix = [&](int kk) -> double2 {  // sum of the wiped samples k0 .. kk-1, kk in (k0, k1)
            const int pos = kk - k0, seg = pos / SEG;
            const double2 b = s_base[seg];
            const PT2 l = s_loc[pos + seg];
            return make_double2(b.x + (double)l.x, b.y + (double)l.y);
        };
        if (MODE != BDS_TRACK_B2A) {
            // B1C: ~21 half-chip steps per replica and pass -- one lane per (replica, step), the three replicas laid
            // end to end (63-66 items: one iteration as a rule), instead of three searches on a third of the lanes
            const int lo = lo1;
            auto at1 = [&](int i1) -> char2 {
                BDS_DASSERT(!use1 || (i1 - lo >= 0 && i1 - lo < kCap1));  // inside the table slice staged in LDS
                return use1 ? s_t1[i1 - lo] : tab2_at(prim_d, NU, i1);
            };
            const int n0 = ub1[0] - ua1[0], n1 = ub1[1] - ua1[1], n2 = ub1[2] - ua1[2];
            for (int r = lane; r < n0 + n1 + n2; r += 64) {
                const int ph = r < n0 ? 0 : r < n0 + n1 ? 1 : 2;
                const int u = (ph == 0 ? ua1[0] + r : ph == 1 ? ua1[1] + (r - n0) : ua1[2] + (r - n0 - n1)) + 1;
                const ColonVec &cl = ph == 0 ? cv3[0] : ph == 1 ? cv3[1] : cv3[2];
                const double thr = (double)(u - 1);
                int kk = (int)floor((thr - cl.a) * inv_inc) + 1;  // a prediction, confirmed with the exact expression
                kk = max(k0 + 1, min(kk, k1 - 1));
                if (!(!(code_arg<0>(cl, kk - 1) > thr) && code_arg<0>(cl, kk) > thr))
                    kk = first_sample_of<0>(cl, inv_inc, u, k0, k1 - 1);
                const double2 S = prefix(kk);
                const char2 cn = at1(u + 1), co = at1(u);
                const double dd = (double)((int)cn.x - (int)co.x), dp = (double)((int)cn.y - (int)co.y);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double d = ph == q ? dd : 0.0;
                    acc[2 * q] = fma(-d, S.x, acc[2 * q]);
                    acc[2 * q + 1] = fma(-d, S.y, acc[2 * q + 1]);
                    if (pilot) {
                        const double e = ph == q ? dp : 0.0;
                        acc[6 + 2 * q] = fma(-e, S.x, acc[6 + 2 * q]);
                        acc[7 + 2 * q] = fma(-e, S.y, acc[7 + 2 * q]);
                    }
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    const char2 cl = at1(ub1[ph] + 1);
                    acc[2 * ph] += (double)cl.x * ti;
                    acc[2 * ph + 1] += (double)cl.x * tq;
                    if (pilot) {
                        acc[6 + 2 * ph] += (double)cl.y * ti;
                        acc[7 + 2 * ph] += (double)cl.y * tq;
                    }
                }
            }
        } else
        {  // code (and pilot code) at the look-up's own resolution: E, P, L steps of one rank together
            const int lo = lo1;
            auto at1 = [&](int i1) -> char2 {
                BDS_DASSERT(!use1 || (i1 - lo >= 0 && i1 - lo < kCap1));  // inside the table slice staged in LDS
                return use1 ? s_t1[i1 - lo] : tab2_at(prim_d, NU, i1);
            };
            const int nmax = max(ub1[0] - ua1[0], max(ub1[1] - ua1[1], ub1[2] - ua1[2]));
            for (int r = lane; r < nmax; r += 64) {
                int kk[3], uu[3];
                bool val[3], bad = false;
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    uu[ph] = ua1[ph] + 1 + r;
                    val[ph] = uu[ph] <= ub1[ph];
                    const double thr = (double)(uu[ph] - 1);
                    int kc = (int)floor((thr - cv3[ph].a) * inv_inc) + 1;  // a prediction, confirmed with the exact expression
                    kc = max(k0 + 1, min(kc, k1 - 1));
                    const bool good = !(code_arg<0>(cv3[ph], kc - 1) > thr) && code_arg<0>(cv3[ph], kc) > thr;
                    bad |= val[ph] && !good;
                    kk[ph] = val[ph] ? kc : k0 + 1;
                }
                if (bad) {
#pragma unroll
                    for (int ph = 0; ph < 3; ++ph)
                        if (val[ph]) kk[ph] = first_sample_of<0>(cv3[ph], inv_inc, uu[ph], k0, k1 - 1);
                }
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    const double2 S = prefix(kk[ph]);
                    const int u = val[ph] ? uu[ph] : ua1[ph];  // (u, u + 1) inside the staged range
                    const char2 cn = at1(u + 1), co = at1(u);
                    const double dd = val[ph] ? (double)((int)cn.x - (int)co.x) : 0.0;
                    acc[2 * ph] = fma(-dd, S.x, acc[2 * ph]);
                    acc[2 * ph + 1] = fma(-dd, S.y, acc[2 * ph + 1]);
                    if (pilot) {
                        const double dp = val[ph] ? (double)((int)cn.y - (int)co.y) : 0.0;
                        acc[6 + 2 * ph] = fma(-dp, S.x, acc[6 + 2 * ph]);
                        acc[7 + 2 * ph] = fma(-dp, S.y, acc[7 + 2 * ph]);
                    }
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    const char2 cl = at1(ub1[ph] + 1);
                    acc[2 * ph] += (double)cl.x * ti;
                    acc[2 * ph + 1] += (double)cl.x * tq;
                    if (pilot) {
                        acc[6 + 2 * ph] += (double)cl.y * ti;
                        acc[7 + 2 * ph] += (double)cl.y * tq;
                    }
                }
            }
        }
        if (MODE == BDS_TRACK_WB && pilot) {  // pilotBOC61(ceil(tcode*6)+1)  (WB_tracking.m:298,311,324)
            const int lo = lo6;
            auto at6 = [&](int i1) -> float {
                BDS_DASSERT(!use6 || (i1 - lo >= 0 && i1 - lo < kCap6));
                return use6 ? (float)s_t6[i1 - lo] : tab_at(prim_p, n6, i1);
            };
            const int nmax = max(ub6[0] - ua6[0], max(ub6[1] - ua6[1], ub6[2] - ua6[2]));
            for (int r = lane; r < nmax; r += 64) {
                int kk[3], uu[3];
                bool val[3], bad = false;
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    uu[ph] = ua6[ph] + 1 + r;
                    val[ph] = uu[ph] <= ub6[ph];
                    const double thr = (double)(uu[ph] - 1);
                    int kc = (int)floor((thr * (1.0 / 6.0) - cv3[ph].a) * inv_inc) + 1;
                    kc = max(k0 + 1, min(kc, k1 - 1));
                    const bool good = !(code_arg<1>(cv3[ph], kc - 1) > thr) && code_arg<1>(cv3[ph], kc) > thr;
                    bad |= val[ph] && !good;
                    kk[ph] = val[ph] ? kc : k0 + 1;
                }
                if (bad) {
#pragma unroll
                    for (int ph = 0; ph < 3; ++ph)
                        if (val[ph]) kk[ph] = first_sample_of<1>(cv3[ph], inv_inc, uu[ph], k0, k1 - 1);
                }
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    const double2 S = prefix(kk[ph]);
                    const int u = val[ph] ? uu[ph] : ua6[ph];
                    const double d6 = val[ph] ? (double)(at6(u + 1) - at6(u)) : 0.0;
                    acc[12 + 2 * ph] = fma(-d6, S.x, acc[12 + 2 * ph]);
                    acc[13 + 2 * ph] = fma(-d6, S.y, acc[13 + 2 * ph]);
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    const double cl = (double)at6(ub6[ph] + 1);
                    acc[12 + 2 * ph] += cl * ti;
                    acc[13 + 2 * ph] += cl * tq;
                }
            }
        }
    }
    // workgroup reduction
    __syncthreads();
    double(*s_part)[kNSums] = reinterpret_cast<double(*)[kNSums]>(trk_lds);
#pragma unroll
    for (int i = 0; i < kNSums; ++i) {
        if (i >= 12 && MODE != BDS_TRACK_WB) {  // no BOC(6,1) correlators outside wide-band mode
            if (lane == 63) s_part[wave][i] = 0.0;
            continue;
        }
        const double v = wave_scan_incl(acc[i]);
        if (lane == 63) s_part[wave][i] = v;
    }
    __syncthreads();
    if (t < kNSums) {
        double v = 0;
        for (int w = 0; w < NW; ++w) v += s_part[w][t];
        sums[t] = v;
    }
}

// grid (nblocks, n_ch); part: [n_ch][nblocks][18]
template <int MODE>
__device__ __forceinline__ void apply_update(const TrkParams &p, ChanState &s, const double *__restrict__ part, int ch,
                                             int nblocks, int epoch, const TrkOut &o, bool store, unsigned char *scratch);

// One kernel per (tracker, correlator variant, record type): a kernel holding all variants ran out of SGPRs
// (150 spilled in wide-band mode, each a lane write + read on the vector unit).
// With `part_prev` the kernel first applies the loop update of the previous epoch: every workgroup of a channel
// recomputes it from that epoch's partial sums (fixed summation order: the same bits everywhere), so no workgroup has to
// wait for another and an epoch is one launch instead of two (the update as its own kernel costs ~5 us, mostly the
// floor of a small dependent launch).  State and partial sums ping-pong between two buffers; workgroup 0 of the channel
// writes the results of the previous epoch and the state the current one starts from.
template <int MODE, int SEG, bool CPLX, int PREC>
__global__ __launch_bounds__(kTrkThreads, BDS_TRK_MINW) void k_trk_correlate(const int8_t *__restrict__ data,
                                                              const int8_t *__restrict__ prim, TrkParams p,
                                                              const ChanState *__restrict__ st_in, ChanState *__restrict__ st_out,
                                                              const double *__restrict__ part_prev, double *__restrict__ part,
                                                              int nblocks, int epoch, const TrkOut *__restrict__ op) {
    const int ch = blockIdx.y;
    ChanState s = st_in[ch];
    // (the table of result arrays comes by pointer: 21 pointers by value cost this kernel 40 SGPRs it does not have)
    if (part_prev && s.active == 1) {
        if constexpr (SEG > 0) {  // the update's scratch sits in the dynamic region the correlator takes over afterwards
            extern __shared__ __attribute__((aligned(16))) unsigned char trk_lds[];
            apply_update<MODE>(p, s, part_prev, ch, nblocks, epoch - 1, *op, blockIdx.x == 0, trk_lds);
        } else {
            __shared__ __attribute__((aligned(16))) unsigned char scratch[kUpdScratch];
            apply_update<MODE>(p, s, part_prev, ch, nblocks, epoch - 1, *op, blockIdx.x == 0, scratch);
        }
    }
    if (st_out != st_in && blockIdx.x == 0 && threadIdx.x == 0) st_out[ch] = s;  // the state this epoch runs with
    double *out = part + ((long)ch * nblocks + blockIdx.x) * kNSums;
    if (s.active != 1) return;
    const EpochGeom g = epoch_geom(s, p);
    const long k0 = (long)blockIdx.x * p.chunk;
    // beyond the block; short read (the update kernel stops the channel); outside the loaded window (it flags it)
    if (k0 >= g.blk || g.pos + g.blk > p.n_bytes || g.pos < p.base || g.pos + g.blk > p.win_end) {
        if (threadIdx.x < kNSums) out[threadIdx.x] = 0.0;
        return;
    }
    const int8_t *pd = prim + ((long)(s.prn - 1) * 2 + 0) * kTabStride;  // (data, pilot) pairs
    const int8_t *pp = prim + ((long)(s.prn - 1) * 2 + 1) * kTabStride;  // pilot BOC(6,1)
    if constexpr (SEG > 0)  // run-based correlator, SEG samples per lane and pass
        correlate_runs<MODE, SEG, CPLX, PREC>(data, pd, pp, p, g, k0, (long)nblocks * p.chunk, p.pilot != 0, out);
    else  // per-sample correlator (reads p.cplx itself)
        correlate_slice<MODE>(data, pd, pp, p, g, k0, (long)nblocks * p.chunk, p.pilot != 0, out);
}

// Open-loop variant: geometry supplied by the caller (bds_track_correlate).
template <int MODE, int SEG, bool CPLX, int PREC>
__global__ __launch_bounds__(kTrkThreads) void k_trk_correlate_open(const int8_t *__restrict__ data,
                                                                   const int8_t *__restrict__ prim, TrkParams p,
                                                                   const int *__restrict__ prn,
                                                                   const double *__restrict__ state6,
                                                                   double *__restrict__ part, int nblocks) {
    const int ch = blockIdx.y;
    const double *s6 = state6 + (long)ch * 6;
    EpochGeom g;
    g.pos = (long long)s6[0];
    g.blk = (long)s6[1];
    g.rem = s6[2];
    g.step = s6[3] / p.fs;
    g.remCarr = s6[4];
    g.carrFreq = s6[5];
    double *out = part + ((long)ch * nblocks + blockIdx.x) * kNSums;
    const long k0 = (long)blockIdx.x * p.chunk;
    if (k0 >= g.blk || g.pos + g.blk > p.n_bytes || g.pos < 0) {
        if (threadIdx.x < kNSums) out[threadIdx.x] = 0.0;
        return;
    }
    const int8_t *pd = prim + ((long)(prn[ch] - 1) * 2 + 0) * kTabStride;  // (data, pilot) pairs
    const int8_t *pp = prim + ((long)(prn[ch] - 1) * 2 + 1) * kTabStride;  // pilot BOC(6,1)
    if constexpr (SEG > 0)  // run-based correlator, SEG samples per lane and pass
        correlate_runs<MODE, SEG, CPLX, PREC>(data, pd, pp, p, g, k0, (long)nblocks * p.chunk, p.pilot != 0, out);
    else  // per-sample correlator (reads p.cplx itself)
        correlate_slice<MODE>(data, pd, pp, p, g, k0, (long)nblocks * p.chunk, p.pilot != 0, out);
}

// launch k<MODE, SEG, CPLX, PREC> for the run-time (mode, runs, cplx, prec) of p; the f64 prefix sums of PREC >= 1 need
// more dynamic LDS than the default limit of a kernel: raised once per kernel and context
#define BDS_TRK_LAUNCH(KERN, grid, lds, stream, ...)                                                        \
    do {                                                                                                    \
        auto fire = [&](auto kern) {                                                                        \
            if ((lds) > 48 * 1024 && ctx->lds_attr_done.insert((const void *)kern).second)                  \
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)); \
            hipLaunchKernelGGL(kern, grid, dim3(kTrkThreads), lds, stream, __VA_ARGS__);                    \
        };                                                                                                  \
        auto go = [&](auto mode_c, auto prec_c) {                                                           \
            constexpr int M = decltype(mode_c)::value, PR = decltype(prec_c)::value;                        \
            if (p.runs == 16 && !p.cplx) fire(KERN<M, 16, false, PR>);                                      \
            else if (p.runs == 16) fire(KERN<M, 16, true, PR>);                                             \
            else if (p.runs == 8 && !p.cplx) fire(KERN<M, 8, false, PR>);                                   \
            else if (p.runs == 8) fire(KERN<M, 8, true, PR>);                                               \
            else fire(KERN<M, 0, false, 0>);                                                                \
        };                                                                                                  \
        auto gp = [&](auto mode_c) {                                                                        \
            if (p.prec == 0) go(mode_c, std::integral_constant<int, 0>{});                                  \
            else if (p.prec == 1) go(mode_c, std::integral_constant<int, 1>{});                             \
            else if (p.prec == 2) go(mode_c, std::integral_constant<int, 2>{});                             \
            else if (p.prec == 3) go(mode_c, std::integral_constant<int, 3>{});                             \
            else if (p.prec == 4) go(mode_c, std::integral_constant<int, 4>{});                             \
            else go(mode_c, std::integral_constant<int, 5>{});                                              \
        };                                                                                                  \
        if (p.mode == BDS_TRACK_B2A) gp(std::integral_constant<int, BDS_TRACK_B2A>{});                      \
        else if (p.mode == BDS_TRACK_NB) gp(std::integral_constant<int, BDS_TRACK_NB>{});                   \
        else gp(std::integral_constant<int, BDS_TRACK_WB>{});                                               \
    } while (0)

__global__ void k_trk_reduce_open(const double *__restrict__ part, int nblocks, double *__restrict__ sums) {
    const int ch = blockIdx.x;
    if (threadIdx.x < kNSums) {
        double v = 0;
        for (int b = 0; b < nblocks; ++b) v += part[((long)ch * nblocks + b) * kNSums + threadIdx.x];
        sums[(long)ch * kNSums + threadIdx.x] = v;
    }
}

// one workgroup per channel: kUpdGroups x 18 threads add up the correlate workgroups' partial sums,
// thread 0 then runs the loop filters
// Loop update of one channel, run by a whole workgroup (>= kUpdGroups * 18 threads): `s` is the state the epoch was
// correlated with, `part` that epoch's partial sums; every thread returns with the next state in `s`.  The result
// arrays of the epoch are written only if `store` (one workgroup per channel).
template <int MODE>
__device__ __forceinline__ void apply_update(const TrkParams &p, ChanState &s, const double *__restrict__ part, int ch,
                                             int nblocks, int epoch, const TrkOut &o, bool store, unsigned char *scratch) {
    // kUpdScratch bytes of LDS (the caller's: at the head of a correlate launch the region the correlator uses afterwards)
    double *s_sum = reinterpret_cast<double *>(scratch);                                     // [18]
    double(*s_grp)[kNSums] = reinterpret_cast<double(*)[kNSums]>(scratch + kNSums * 8);      // [kUpdGroups][18]
    double *s_x = reinterpret_cast<double *>(scratch + (1 + kUpdGroups) * kNSums * 8);       // [6]
    ChanState &s_next = *reinterpret_cast<ChanState *>(scratch + (1 + kUpdGroups) * kNSums * 8 + 48);
    const EpochGeom g = epoch_geom(s, p);
    // fixed-order two-level sum over the correlate workgroups (bit-stable from run to run): group j
    // takes workgroups j, j+14, ...; the 14 group sums are then added in order.  (A single thread per
    // sum walking all ~243 workgroups of a 10-ms epoch cost 60 us of dependent loads per epoch.)
    if (threadIdx.x < kUpdGroups * kNSums) {
        const int grp = threadIdx.x / kNSums, i = threadIdx.x - grp * kNSums;
        const long nb = min((long)nblocks, (g.blk + p.chunk - 1) / p.chunk);
        double v = 0;
        for (long b = grp; b < nb; b += kUpdGroups) v += part[((long)ch * nblocks + b) * kNSums + i];
        s_grp[grp][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double v = 0;
#pragma unroll
        for (int j = 0; j < kUpdGroups; ++j) v += s_grp[j][threadIdx.x];
        s_sum[threadIdx.x] = v;
    }
    __syncthreads();
    const double two_pi = 6.283185307179586476925286766559;
    const double pi = 3.14159265358979323846;
    const double L = (double)p.code_len;
    const double I_E = s_sum[0], Q_E = s_sum[1], I_P = s_sum[2], Q_P = s_sum[3], I_L = s_sum[4], Q_L = s_sum[5];
    double pI_E = s_sum[6], pQ_E = s_sum[7], pI_P = s_sum[8], pQ_P = s_sum[9], pI_L = s_sum[10], pQ_L = s_sum[11];
    double cI_E = 0, cQ_E = 0, cI_P = 0, cQ_P = 0, cI_L = 0, cQ_L = 0;
    if (MODE == BDS_TRACK_WB && p.pilot) {  // QMBOC composite, WB_tracking.m:375-380
        const double a = sqrt(4.0 / 33), b = sqrt(29.0 / 33);
        cI_E = -a * s_sum[12] + b * pQ_E;
        cQ_E = -a * s_sum[13] - b * pI_E;
        cI_P = -a * s_sum[14] + b * pQ_P;
        cQ_P = -a * s_sum[15] - b * pI_P;
        cI_L = -a * s_sum[16] + b * pQ_L;
        cQ_L = -a * s_sum[17] - b * pI_L;
    }
    // The update is a chain of dependent f64 operations (two atan, up to six sqrt, an fmod, divisions: ~3 us on one
    // thread, and at the head of a correlate launch every workgroup waits for it).  Its independent pieces run on four
    // waves of the workgroup (different SIMDs), one lane each, with exactly the expressions of the serial form; thread 0
    // then combines them in the reference's order.
    // s_x: data atan, pilot atan, data (E-L)/(E+L), pilot (E-L)/(E+L), next code phase, next carrier phase
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) {
        const int w = threadIdx.x >> 6;
        if (w == 0) {
            s_x[0] = atan(Q_P / I_P) / two_pi;  // :337 (true division: I = 0 -> +-pi/2, 0/0 -> NaN, as MATLAB)
        } else if (w == 1) {
            double cq = 0.0;
            if (p.pilot) {
                if (MODE == BDS_TRACK_B2A) {
                    // QI = (pI + j pQ) * exp(-j pi/2); atan(imag/real)  (:345-348)
                    const double cr = cos(-pi / 2), sr = sin(-pi / 2);  // exp(-1i*pi/2) as MATLAB evaluates it
                    const double re = pI_P * cr - pQ_P * sr, im = pI_P * sr + pQ_P * cr;
                    cq = atan(im / re) / two_pi;
                } else if (MODE == BDS_TRACK_NB) {
                    cq = atan(-pI_P / pQ_P) / two_pi;  // NB:357
                } else {
                    cq = atan(cQ_P / cI_P) / two_pi;  // WB:392
                }
            }
            s_x[1] = cq;
        } else if (w == 2) {
            const double eE = sqrt(I_E * I_E + Q_E * Q_E), eL = sqrt(I_L * I_L + Q_L * Q_L);
            s_x[2] = (eE - eL) / (eE + eL);  // :366
            double pce = 0.0;
            if (p.pilot) {
                double pE, pL;
                if (MODE == BDS_TRACK_WB) {
                    pE = sqrt(cI_E * cI_E + cQ_E * cQ_E);
                    pL = sqrt(cI_L * cI_L + cQ_L * cQ_L);
                } else {
                    pE = sqrt(pI_E * pI_E + pQ_E * pQ_E);
                    pL = sqrt(pI_L * pI_L + pQ_L * pQ_L);
                }
                pce = (pE - pL) / (pE + pL);
            }
            s_x[3] = pce;
        } else {
            // remCodePhase = tcode(blksize) + codePhaseStep - codeLength  (:295; B1C: tcode/2, WB:327)
            if (MODE == BDS_TRACK_B2A) {
                const double tp_last = g.rem + (double)(g.blk - 1) * g.step;
                s_x[4] = (tp_last + g.step) - L;
            } else {
                const double tp_last = g.rem * 2.0 + (double)(g.blk - 1) * (g.step * 2.0);
                s_x[4] = tp_last / 2 + g.step - L;
            }
            // remCarrPhase = rem(trigarg(blksize+1), 2*pi)  (:303-305)
            const double t_end = (double)g.blk / p.fs;
            const double trig_end = ((s.carrFreq * 2.0 * pi) * t_end) + s.remCarrPhase;
            s_x[5] = fmod(trig_end, two_pi);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        [&] {
    const long e = (long)ch * p.n_epochs + epoch;
    if (g.pos + g.blk <= p.n_bytes && (g.pos < p.base || g.pos + g.blk > p.win_end)) {
        // inside the file but outside the part of it that was loaded: nothing of this epoch is valid;
        // the host repeats the call with the whole record in HBM
        s.active = -2;
        return;
    }
    if (store) o.absoluteSample[e] = (double)s.pos;  // tracking.m:226 (assigned before the read)
    if (g.pos + g.blk > p.n_bytes) {
        // short read: message + return in the reference (tracking.m:250-254); partial results stay
        s.active = 0;
        return;
    }
    if (store) o.remCodePhase[e] = s.remCodePhase;  // :258
    if (store) o.remCarrPhase[e] = s.remCarrPhase;  // :300
    const double rem_next = s_x[4], carr_next = s_x[5];
    // ---- PLL discriminator ----------------------------------------------------------------------
    double carrError = s_x[0];
    if (p.pilot) {
        const double cq = s_x[1];
        if (MODE == BDS_TRACK_B2A)
            carrError = (carrError + cq) / 2;  // :352
        else if (MODE == BDS_TRACK_NB)
            carrError = (carrError * 11 + cq * 29) / 40;  // NB:360
        else
            carrError = (carrError * 1 + cq * 3) / 4;  // WB:395
    }
    s.d2CarrError = s.d2CarrError + carrError * p.pf3;                    // :356
    s.dCarrError = s.d2CarrError + carrError * p.pf2 + s.dCarrError;      // :357
    const double carrNco = s.dCarrError + carrError * p.pf1;              // :358
    if (store) o.carrFreq[e] = s.carrFreq;                                           // :361
    // ---- DLL discriminator ---------------------------------------------------------------------
    double codeError = s_x[2];  // :366
    if (MODE != BDS_TRACK_B2A) codeError = codeError * (1 - p.spacing);  // WB:409-410
    if (p.pilot) {
        const double pce = s_x[3];
        if (MODE == BDS_TRACK_B2A)
            codeError = (codeError + pce) / 2;  // :377
        else if (MODE == BDS_TRACK_NB)
            codeError = (codeError * 11 + pce * (1 - p.spacing) * 29) / 40;  // NB:381-384
        else
            codeError = codeError * p.factor + pce * (1 - p.spacing) * (1 - p.factor);  // WB:418
    }
    const double codeNco = s.oldCodeNco + (p.tau2 / p.tau1) * (codeError - s.oldCodeError) + codeError * (p.pdi / p.tau1);  // :381
    s.oldCodeNco = codeNco;
    s.oldCodeError = codeError;
    if (store) o.codeFreq[e] = s.codeFreq;  // :387
    if (store) o.dllDiscr[e] = codeError;
    if (store) o.dllDiscrFilt[e] = codeNco;
    if (store) o.pllDiscr[e] = carrError;
    if (store) o.pllDiscrFilt[e] = carrNco;
    if (store) o.I_E[e] = I_E;
    if (store) o.I_P[e] = I_P;
    if (store) o.I_L[e] = I_L;
    if (store) o.Q_E[e] = Q_E;
    if (store) o.Q_P[e] = Q_P;
    if (store) o.Q_L[e] = Q_L;
    if (p.pilot) {
        if (MODE == BDS_TRACK_WB) {
            if (store) o.Pilot_I_E[e] = cI_E;
            if (store) o.Pilot_Q_E[e] = cQ_E;
            if (store) o.Pilot_I_P[e] = cI_P;
            if (store) o.Pilot_Q_P[e] = cQ_P;
            if (store) o.Pilot_I_L[e] = cI_L;
            if (store) o.Pilot_Q_L[e] = cQ_L;
        } else {
            if (store) o.Pilot_I_P[e] = pI_P;
            if (store) o.Pilot_Q_P[e] = pQ_P;
        }
    }
    s.carrFreq = s.carrFreqBasis + carrNco;   // :363
    s.codeFreq = s.codeFreqBasis - codeNco;   // :389
    s.remCodePhase = rem_next;
    s.remCarrPhase = carr_next;
    s.pos += g.blk;
    s.completed = epoch + 1;
        }();
        s_next = s;
    }
    __syncthreads();
    s = s_next;
    __syncthreads();  // the scratch may be reused from here on
}

template <int MODE>
__global__ __launch_bounds__(kUpdThreads) void k_trk_update(TrkParams p, ChanState *__restrict__ st,
                                                   const double *__restrict__ part, int nblocks, int epoch,
                                                   TrkOut o) {
    const int ch = blockIdx.x;
    __shared__ __attribute__((aligned(16))) unsigned char scratch[kUpdScratch];
    ChanState s = st[ch];
    if (s.active != 1) return;
    apply_update<MODE>(p, s, part, ch, nblocks, epoch, o, true, scratch);
    if (threadIdx.x == 0) st[ch] = s;
}

struct TrackState {
    int8_t *d_data = nullptr;
    size_t data_cap = 0;
    size_t loaded_bytes = 0;  // bytes of the record the last call copied to HBM (diagnostics)
    int8_t *d_prim = nullptr;
    int prim_signal = 0;
};

void track_state_free(TrackState *t) {
    if (!t) return;
    if (t->d_data) (void)hipFree(t->d_data);
    if (t->d_prim) (void)hipFree(t->d_prim);
    delete t;
}

static hipStream_t st(bds_ctx *ctx) { return (hipStream_t)ctx->stream; }

static int track_mode(const bds_settings &s) {
    if (s.signal == BDS_SIGNAL_B2A) return BDS_TRACK_B2A;
    return s.pilotTRKflag == 2 ? BDS_TRACK_WB : BDS_TRACK_NB;  // B1C/postProcessing.m:137-143
}

static int pilot_on(const bds_settings &s, int mode) {
    if (mode == BDS_TRACK_WB) return s.pilotTRKflag == 2;
    return s.pilotTRKflag == 1;  // tracking.m:70, NB_tracking.m:78
}

static int ensure_prim(bds_ctx *ctx, TrackState &t, int signal) {
    if (t.d_prim && t.prim_signal == signal) return BDS_OK;
    // per PRN two slots: (data, pilot) pairs at the code's own resolution, and the pilot BOC(6,1)
    // array; wrapped padding on both sides (tab_at / tab2_at)
    const size_t bytes = (size_t)BDS_MAX_PRN * 2 * kTabStride;
    if (!t.d_prim) BDS_HIP(ctx, hipMalloc((void **)&t.d_prim, bytes));
    std::vector<int8_t> tab(bytes, 0);
    int8_t prim[2][10230];
    for (int prn = 1; prn <= BDS_MAX_PRN; ++prn) {
        gen_primary(signal, false, prn, prim[0]);
        gen_primary(signal, true, prn, prim[1]);
        auto unit = [&](int comp, int units, long j) {  // value at padded position j of a `units`-per-chip array
            const long n = 10230L * units;
            long u = (j - kTabPad) % n;
            if (u < 0) u += n;
            const int8_t chip = prim[comp][u / units];
            const long sub = u % units;  // BOC(1,1): [-c, +c]; BOC(6,1): (-1)^ii c, ii = sub + 1
            return units == 1 ? chip : ((sub & 1) ? chip : (int8_t)-chip);
        };
        const int units = signal == BDS_SIGNAL_B1C ? 2 : 1;
        int8_t *dp = &tab[((size_t)(prn - 1) * 2 + 0) * kTabStride];
        for (long j = 0; j < 10230L * units + 2 * kTabPad; ++j) {
            dp[2 * j] = unit(0, units, j);
            dp[2 * j + 1] = unit(1, units, j);
        }
        if (signal == BDS_SIGNAL_B1C) {
            int8_t *b6 = &tab[((size_t)(prn - 1) * 2 + 1) * kTabStride];
            for (long j = 0; j < 10230L * 12 + 2 * kTabPad; ++j) b6[j] = unit(1, 12, j);
        }
    }
    BDS_HIP(ctx, hipMemcpy(t.d_prim, tab.data(), bytes, hipMemcpyHostToDevice));
    t.prim_signal = signal;
    return BDS_OK;
}

static int fill_params(bds_ctx *ctx, const bds_settings &s, TrkParams &p, int n_epochs, size_t n_bytes) {
    if (s.signal != BDS_SIGNAL_B1C && s.signal != BDS_SIGNAL_B2A) return fail(ctx, BDS_ERR_ARG, "settings.signal invalid");
    if (s.fileType != 1 && s.fileType != 2) return fail(ctx, BDS_ERR_ARG, "settings.fileType must be 1 (real) or 2 (I/Q)");
    if (s.dataType != 0)  // fread(fid, ..., settings.dataType), tracking.m:237-238
        return fail(ctx, BDS_ERR_UNSUPPORTED, "settings.dataType: only 'schar' (int8 samples) is supported");
    if (s.codeLength != 10230 || !(s.samplingFreq > 0) || !(s.intTime > 0))
        return fail(ctx, BDS_ERR_ARG, "settings.codeLength/samplingFreq/intTime invalid");
    p.mode = track_mode(s);
    p.pilot = pilot_on(s, p.mode);
    p.code_len = s.codeLength;
    p.n_epochs = n_epochs;
    p.fs = s.samplingFreq;
    p.inv_fs = 1.0 / s.samplingFreq;
    p.spacing = s.dllCorrelatorSpacing;
    bds_calc_loop_coef(s.dllNoiseBandwidth, s.dllDampingRatio, 1.0, &p.tau1, &p.tau2);  // tracking.m:110-112
    bds_calc_loop_coef_carr(&s, &p.pf3, &p.pf2, &p.pf1);                                 // :116
    p.pdi = s.intTime;                                                                   // :107
    p.factor = p.mode == BDS_TRACK_WB ? bds_calc_weighing_factor(&s) : 0.0;              // WB_tracking.m:138
    p.cplx = s.fileType == 2;
    if (ctx->tune.trk_persample) {  // per-sample correlator (the round-1 kernel; A/B and cross-check)
        p.chunk = s.signal == BDS_SIGNAL_B2A ? 2048 : 8192;
        if (ctx->tune.trk_chunk > 0) p.chunk = std::max(256, ctx->tune.trk_chunk);
        p.runs = 0;
    } else {  // run-based correlator: 8 or 16 consecutive samples per thread
        p.runs = s.signal == BDS_SIGNAL_B2A ? 8 : 16;
        if (ctx->tune.trk_chunk == 2048) p.runs = 8;
        if (ctx->tune.trk_chunk == 4096) p.runs = 16;
        p.chunk = kTrkThreads * p.runs;
        p.prec = std::max(0, std::min(5, ctx->tune.trk_prec));
        if (ctx->tune.trk_seg == 8 || ctx->tune.trk_seg == 16) p.runs = ctx->tune.trk_seg, p.chunk = kTrkThreads * p.runs;
    }
    p.n_bytes = (long long)(n_bytes / (p.cplx ? 2 : 1));  // whole samples an fread can deliver
    return BDS_OK;
}

template <class F>
static void for_each_field(bds_track_out *o, TrkOut *d, F f) {
    f(o->absoluteSample, d->absoluteSample, 0.0);
    f(o->codeFreq, d->codeFreq, INFINITY);
    f(o->carrFreq, d->carrFreq, INFINITY);
    f(o->I_P, d->I_P, 0.0);
    f(o->I_E, d->I_E, 0.0);
    f(o->I_L, d->I_L, 0.0);
    f(o->Q_E, d->Q_E, 0.0);
    f(o->Q_P, d->Q_P, 0.0);
    f(o->Q_L, d->Q_L, 0.0);
    f(o->Pilot_I_P, d->Pilot_I_P, 0.0);
    f(o->Pilot_Q_P, d->Pilot_Q_P, 0.0);
    f(o->Pilot_I_E, d->Pilot_I_E, 0.0);
    f(o->Pilot_I_L, d->Pilot_I_L, 0.0);
    f(o->Pilot_Q_E, d->Pilot_Q_E, 0.0);
    f(o->Pilot_Q_L, d->Pilot_Q_L, 0.0);
    f(o->dllDiscr, d->dllDiscr, INFINITY);
    f(o->dllDiscrFilt, d->dllDiscrFilt, INFINITY);
    f(o->pllDiscr, d->pllDiscr, INFINITY);
    f(o->pllDiscrFilt, d->pllDiscrFilt, INFINITY);
    f(o->remCodePhase, d->remCodePhase, INFINITY);
    f(o->remCarrPhase, d->remCarrPhase, INFINITY);
}

// include/Calc_CNo_PLD.m (B1C :45-114, B2a :38-100) over prompt values [k-M, k)
__device__ static void cno_pld_one(const double *I, const double *Q, int M, double T, double *lin, double *cno, double *pld) {
    double zm = 0;
    for (int i = 0; i < M; ++i) zm += I[i] * I[i] + Q[i] * Q[i];
    zm /= M;
    double zv = 0;
    for (int i = 0; i < M; ++i) {
        const double z = I[i] * I[i] + Q[i] * Q[i];
        zv += (z - zm) * (z - zm);
    }
    zv /= (M - 1);  // var(): N-1
    const double pav = sqrt(zm * zm - zv);
    const double nv = 0.5 * (zm - pav);
    *lin = fabs((1 / T) * pav / (2 * nv));
    *cno = 10 * log10(*lin);
    double sp = 0, sn = 0, sq = 0;
    for (int i = 0; i < M; ++i) {
        if (I[i] > 0) sp += I[i];
        if (I[i] < 0) sn += I[i];
        sq += Q[i];
    }
    const double a = (sp - sn) * (sp - sn);
    *pld = (a - sq * sq) / (a + sq * sq);
}

// C/N0 + phase-lock detector of every finished CNoInterval (tracking.m:411-434), on the device arrays the
// tracking loop just wrote: one workgroup per channel, one thread per interval, then the reference's
// two-point smoothing CNo = new/2 + previous/2 (:420-421; previous = 0 before the first interval).
// cno5: [5][n_ch][n_cno] = DataCNo, DataPLD, PilotCNo, PilotPLD, SigCNo; pm: 0 no pilot, 1 pilot with
// I/Q swapped (narrow-band, Calc_CNo_PLD.m:84-87), 2 pilot as is (wide-band).
__global__ __launch_bounds__(256) void k_trk_cno(TrkOut o, const ChanState *__restrict__ st, int n_epochs, int M,
                                                 int n_cno, int pm, double T, double *__restrict__ raw3,
                                                 double *__restrict__ cno5) {
    const int ch = blockIdx.x, n_ch = gridDim.x;
    const int done = st[ch].prn ? st[ch].completed : 0;
    const size_t plane = (size_t)n_ch * n_cno;
    double *r = raw3 + (size_t)ch * n_cno * 3;
    for (int q = threadIdx.x; q < n_cno; q += blockDim.x) {
        const size_t e = (size_t)ch * n_cno + q;
        for (int f = 0; f < 5; ++f) cno5[f * plane + e] = 0;
        r[q * 3 + 0] = r[q * 3 + 1] = r[q * 3 + 2] = 0;
        if ((q + 1) * M > done) continue;
        const size_t o0 = (size_t)ch * n_epochs + (size_t)q * M;
        double dlin, dcno, dpld, plin = 0, pcno = 0, ppld = 0;
        cno_pld_one(o.I_P + o0, o.Q_P + o0, M, T, &dlin, &dcno, &dpld);
        if (pm == 2)
            cno_pld_one(o.Pilot_I_P + o0, o.Pilot_Q_P + o0, M, T, &plin, &pcno, &ppld);
        else if (pm == 1)
            cno_pld_one(o.Pilot_Q_P + o0, o.Pilot_I_P + o0, M, T, &plin, &pcno, &ppld);
        r[q * 3 + 0] = dcno;
        r[q * 3 + 1] = pcno;
        r[q * 3 + 2] = 10 * log10(dlin + plin);
        cno5[1 * plane + e] = dpld;
        if (pm) cno5[3 * plane + e] = ppld;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < n_cno; q += blockDim.x) {
        if ((q + 1) * M > done) continue;
        const size_t e = (size_t)ch * n_cno + q;
        const double p0 = q ? r[(q - 1) * 3 + 0] : 0.0, p1 = q ? r[(q - 1) * 3 + 1] : 0.0, p2 = q ? r[(q - 1) * 3 + 2] : 0.0;
        cno5[0 * plane + e] = r[q * 3 + 0] * 0.5 + p0 * 0.5;
        if (pm) {
            cno5[2 * plane + e] = r[q * 3 + 1] * 0.5 + p1 * 0.5;
            cno5[4 * plane + e] = r[q * 3 + 2] * 0.5 + p2 * 0.5;
        }
    }
}

// Source of the IF record: copies bytes [off, off + n) of the file into device memory at dst.
using RecordLoader = std::function<int(size_t off, size_t n, int8_t *dst)>;

// whole_file: load every byte (second attempt after a channel's reads left the window of the first)
static int do_track(bds_ctx *ctx, const bds_settings *s, const RecordLoader &load, size_t n_bytes, int n_ch,
                    const bds_channel *channel, bds_track_out *out, bool whole_file = false) {
    if (!ctx || !s || !channel || !out) return BDS_ERR_ARG;
    if (n_ch < 1 || out->n_ch != n_ch || out->n_epochs < 1) return fail(ctx, BDS_ERR_ARG, "bds_track: n_ch / n_epochs mismatch");
    if (!out->completed || !out->status || !out->absoluteSample || !out->I_P || !out->Q_P)
        return fail(ctx, BDS_ERR_ARG, "bds_track: required output arrays missing");
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->trk) ctx->trk = new TrackState();
    TrackState &t = *ctx->trk;
    TrkParams p{};
    const int n_epochs = out->n_epochs;
    int rc = fill_params(ctx, *s, p, n_epochs, n_bytes);
    if (rc) return rc;
    if ((rc = ensure_prim(ctx, t, s->signal))) return rc;
    // channel state (tracking.m:170-188)
    std::vector<ChanState> hs((size_t)n_ch);
    const double max_step_inv = 0;
    (void)max_step_inv;
    double min_code_freq = 1e300;
    for (int c = 0; c < n_ch; ++c) {
        ChanState &cs = hs[c];
        memset(&cs, 0, sizeof(cs));
        cs.prn = channel[c].PRN;
        out->completed[c] = 0;
        out->status[c] = '-';
        if (cs.prn == 0) continue;
        if (cs.prn < 1 || cs.prn > BDS_MAX_PRN) return fail(ctx, BDS_ERR_ARG, "channel(%d).PRN = %d out of range", c + 1, cs.prn);
        cs.active = 1;
        cs.codeFreq = cs.codeFreqBasis = channel[c].codeFreq;
        cs.carrFreq = cs.carrFreqBasis = channel[c].acquiredFreq;
        cs.pos = (long long)s->skipNumberOfBytes + (long long)channel[c].codePhase - 1;  // fseek, :151-153
        if (cs.pos < 0) return fail(ctx, BDS_ERR_ARG, "channel(%d): negative file offset", c + 1);
        if (!(cs.codeFreq > 0)) return fail(ctx, BDS_ERR_ARG, "channel(%d).codeFreq must be > 0", c + 1);
        min_code_freq = std::min(min_code_freq, cs.codeFreq);
    }
    // correlate grid: blksize stays near codeLength*fs/codeFreq; sized for code rates down to 2 % below the slowest
    // channel's (a longer block is walked by the same workgroups in further strides)
    long max_blk = (long)std::ceil((double)s->codeLength / ((min_code_freq < 1e299 ? min_code_freq : s->codeFreqBasis) * 0.98 / s->samplingFreq)) + 2;
    int nblocks = (int)((max_blk + p.chunk - 1) / p.chunk);
    if (p.runs && n_ch > 0) {
        // run-based correlator: every wave pays a fixed cost per launch (tables, first carrier, the 18-sum reduction), so
        // the grid is sized to fill the chip about once (3 workgroups per CU) and each wave walks several passes
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        const int want = std::max(1, 3 * cus / n_ch);
        if (nblocks > want) {
            const int passes = (nblocks + want - 1) / want;
            nblocks = (nblocks + passes - 1) / passes;
        }
    }
    if (ctx->tune.trk_nblocks > 0) nblocks = ctx->tune.trk_nblocks;
    // Only the part of the record the channels can touch goes to HBM: from the earliest start sample to the latest
    // start + n_epochs blocks at a code rate 2 % low (the reference streams blksize samples per epoch with fread,
    // tracking.m:237-240; a recording is usually far longer than msToProcess).  End-of-file is still judged against
    // the real file size (p.n_bytes).
    {
        const long long coeff = p.cplx ? 2 : 1;
        long long first = p.n_bytes, last = 0;
        for (int c = 0; c < n_ch; ++c) {
            if (!hs[c].active) continue;
            const long blk_c = (long)std::ceil((double)s->codeLength / (hs[c].codeFreq * 0.98 / s->samplingFreq)) + 2;
            first = std::min(first, hs[c].pos);
            last = std::max(last, hs[c].pos + (long long)n_epochs * blk_c);
        }
        if (whole_file || first > last) first = 0, last = p.n_bytes;
        first = std::max(0LL, std::min(first, p.n_bytes));
        last = std::max(first, std::min(last, p.n_bytes));
        p.base = first;
        p.win_end = last;
        const size_t wbytes = (size_t)((last - first) * coeff);
        if (t.data_cap < wbytes || !t.d_data) {
            if (t.d_data) (void)hipFree(t.d_data), t.d_data = nullptr, t.data_cap = 0;
            hipError_t e = hipMalloc((void **)&t.d_data, wbytes + kDataSlack);
            if (e != hipSuccess)
                return fail(ctx, BDS_ERR_NOMEM, "IF record window of %zu bytes does not fit in HBM: %s", wbytes, hipGetErrorString(e));
            t.data_cap = std::max<size_t>(wbytes, 1);
        }
        if (wbytes && (rc = load((size_t)(first * coeff), wbytes, t.d_data))) return rc;
        t.loaded_bytes = wbytes;
    }
    // per-call device buffers, released on every exit path
    struct DevScope {
        std::vector<void *> p;
        std::vector<hipEvent_t> ev;
        void add(void *q) { p.push_back(q); }
        void release() {
            for (void *q : p)
                if (q) (void)hipFree(q);
            for (hipEvent_t e : ev) (void)hipEventDestroy(e);
            p.clear();
            ev.clear();
        }
        ~DevScope() { release(); }
    } scope;
    // state and partial sums ping-pong between two buffers when the loop update rides at the head of the next epoch's
    // correlate launch (default); with BDS_TRK_NOFUSE_UPDATE both halves are the same buffer and the update is its own launch
    const bool fuse = !ctx->tune.trk_nofuse_update;
    ChanState *d_st = nullptr;
    double *d_part = nullptr;
    BDS_HIP(ctx, hipMalloc((void **)&d_st, sizeof(ChanState) * n_ch * 2));
    scope.add(d_st);
    const size_t part_n = (size_t)n_ch * nblocks * kNSums;
    BDS_HIP(ctx, hipMalloc((void **)&d_part, sizeof(double) * part_n * 2));
    scope.add(d_part);
    BDS_HIP(ctx, hipMemcpyAsync(d_st, hs.data(), sizeof(ChanState) * n_ch, hipMemcpyHostToDevice, st(ctx)));
    // device result arrays, initialised like the reference template (tracking.m:48-82)
    const size_t ne = (size_t)n_ch * n_epochs;
    TrkOut d{};
    std::vector<double *> allocs;
    std::vector<double> init(ne);
    int err = BDS_OK;
    for_each_field(out, &d, [&](double *host, double *&dev, double v0) {
        if (err) return;
        // arrays the variant does not create still need a device target when the kernel writes them
        hipError_t e = hipMalloc((void **)&dev, sizeof(double) * ne);
        if (e != hipSuccess) {
            err = fail(ctx, BDS_ERR_NOMEM, "hipMalloc track output: %s", hipGetErrorString(e));
            return;
        }
        allocs.push_back(dev);
        scope.add(dev);
        std::fill(init.begin(), init.end(), v0);
        (void)hipMemcpyAsync(dev, init.data(), sizeof(double) * ne, hipMemcpyHostToDevice, st(ctx));
        (void)hipStreamSynchronize(st(ctx));
        (void)host;
    });
    if (err) return err;

    hipEvent_t ev0, ev1;  // (owned by the scope: every early return below destroys them)
    BDS_HIP(ctx, hipEventCreate(&ev0));
    scope.ev.push_back(ev0);
    BDS_HIP(ctx, hipEventCreate(&ev1));
    scope.ev.push_back(ev1);
    BDS_HIP(ctx, hipEventRecord(ev0, st(ctx)));
    const int8_t *data = t.d_data;
    dim3 gc(nblocks, n_ch);
    TrkOut *d_out = nullptr;  // the table of result arrays, for the correlate launches that carry the previous epoch's update
    BDS_HIP(ctx, hipMalloc((void **)&d_out, sizeof(TrkOut)));
    scope.add(d_out);
    BDS_HIP(ctx, hipMemcpyAsync(d_out, &d, sizeof(TrkOut), hipMemcpyHostToDevice, st(ctx)));
    auto launch_update = [&](ChanState *stp, const double *partp, int k) {
        switch (p.mode) {
            case BDS_TRACK_B2A:
                hipLaunchKernelGGL(k_trk_update<BDS_TRACK_B2A>, dim3(n_ch), dim3(kUpdThreads), 0, st(ctx), p, stp, partp, nblocks, k, d);
                break;
            case BDS_TRACK_NB:
                hipLaunchKernelGGL(k_trk_update<BDS_TRACK_NB>, dim3(n_ch), dim3(kUpdThreads), 0, st(ctx), p, stp, partp, nblocks, k, d);
                break;
            default:
                hipLaunchKernelGGL(k_trk_update<BDS_TRACK_WB>, dim3(n_ch), dim3(kUpdThreads), 0, st(ctx), p, stp, partp, nblocks, k, d);
                break;
        }
    };
    ChanState *st_final = d_st;  // where the state ends up
    RoctxRange rg_epochs("trk.epoch_loop");
    for (int k = 0; k < n_epochs; ++k) {
        if (fuse) {
            const int cur = k & 1;
            ChanState *st_in = d_st + (size_t)cur * n_ch, *st_out = d_st + (size_t)(cur ^ 1) * n_ch;
            const double *part_prev = k > 0 ? d_part + (size_t)(cur ^ 1) * part_n : nullptr;
            double *part_cur = d_part + (size_t)cur * part_n;
            BDS_TRK_LAUNCH(k_trk_correlate, gc, (p.runs ? runs_lds_bytes(p.runs, p.prec) : 0), st(ctx), data, (const int8_t *)t.d_prim, p,
                           (const ChanState *)st_in, st_out, part_prev, part_cur, nblocks, k, (const TrkOut *)d_out);
            st_final = st_out;
            if (k == n_epochs - 1) launch_update(st_final, part_cur, k);  // the last epoch's update has no next launch to ride on
        } else {
            BDS_TRK_LAUNCH(k_trk_correlate, gc, (p.runs ? runs_lds_bytes(p.runs, p.prec) : 0), st(ctx), data, (const int8_t *)t.d_prim, p,
                           (const ChanState *)d_st, d_st, (const double *)nullptr, d_part, nblocks, k, (const TrkOut *)d_out);
            launch_update(d_st, d_part, k);
        }
    }
    BDS_HIP(ctx, hipGetLastError());
    BDS_HIP(ctx, hipEventRecord(ev1, st(ctx)));
    BDS_HIP(ctx, hipMemcpyAsync(hs.data(), st_final, sizeof(ChanState) * n_ch, hipMemcpyDeviceToHost, st(ctx)));
    BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));
    float ms = 0;
    BDS_HIP(ctx, hipEventElapsedTime(&ms, ev0, ev1));
    memset(&ctx->timing, 0, sizeof(ctx->timing));
    ctx->timing.total_ms = ms;
    ctx->timing.n_pairs = n_epochs;
    for (int c = 0; c < n_ch; ++c)
        if (hs[c].active == -2) {  // a channel read past the loaded window (code rate > 2 % low): redo with the whole record
            if (whole_file) return fail(ctx, BDS_ERR_HIP, "bds_track: channel %d left the loaded record", c + 1);
            scope.release();  // the retry allocates its own result arrays: do not hold this attempt's beside them
            return do_track(ctx, s, load, n_bytes, n_ch, channel, out, true);
        }

    // The reference tracks channels one after another and returns at the first short read
    // (tracking.m:250-254): that channel keeps its partial results, later ones are never started.
    int first_abort = n_ch;
    for (int c = 0; c < n_ch; ++c)
        if (hs[c].prn != 0 && hs[c].completed < n_epochs) {
            first_abort = c;
            break;
        }
    for_each_field(out, &d, [&](double *host, double *&dev, double v0) {
        if (!host) return;
        (void)hipMemcpy(host, dev, sizeof(double) * ne, hipMemcpyDeviceToHost);
        for (int c = first_abort + 1; c < n_ch; ++c)
            std::fill(host + (size_t)c * n_epochs, host + (size_t)(c + 1) * n_epochs, v0);
        // epochs after the abort stay at the template value (device never wrote them), except
        // absoluteSample of the aborted epoch which the reference assigns before the read
    });
    for (int c = 0; c < n_ch; ++c) {
        if (hs[c].prn == 0 || c > first_abort) {
            out->completed[c] = 0;
            continue;
        }
        out->completed[c] = hs[c].completed;
        if (hs[c].completed == n_epochs) out->status[c] = channel[c].status;  // :441
    }
    // C/N0 + lock detector (tracking.m:411-434): computed on the device from the arrays above
    const int M = s->CNoInterval;
    if (M > 1 && out->n_cno > 0 && out->DataCNo) {
        const int pm = p.pilot ? (p.mode == BDS_TRACK_WB ? 2 : 1) : 0;
        const int nq = out->n_cno;
        double *d_raw = nullptr, *d_cno = nullptr;
        BDS_HIP(ctx, hipMalloc((void **)&d_raw, sizeof(double) * (size_t)n_ch * nq * 3));
        scope.add(d_raw);
        BDS_HIP(ctx, hipMalloc((void **)&d_cno, sizeof(double) * (size_t)n_ch * nq * 5));
        scope.add(d_cno);
        BDS_HIP(ctx, hipMemcpyAsync(d_st, hs.data(), sizeof(ChanState) * n_ch, hipMemcpyHostToDevice, st(ctx)));
        hipLaunchKernelGGL(k_trk_cno, dim3(n_ch), dim3(256), 0, st(ctx), d, (const ChanState *)d_st, n_epochs, M, nq, pm,
                           s->intTime, d_raw, d_cno);
        std::vector<double> h((size_t)n_ch * nq * 5);
        BDS_HIP(ctx, hipMemcpyAsync(h.data(), d_cno, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st(ctx)));
        BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));
        double *dst[5] = {out->DataCNo, out->DataPLD, out->PilotCNo, out->PilotPLD, out->SigCNo};
        const size_t plane = (size_t)n_ch * nq;
        for (int f = 0; f < 5; ++f) {
            if (!dst[f]) continue;
            for (int c = 0; c < n_ch; ++c)
                for (int q = 0; q < nq; ++q) {
                    const bool live = out->completed[c] > 0 && (q + 1) * M <= out->completed[c] && (pm || f < 2);
                    dst[f][(size_t)c * nq + q] = live ? h[f * plane + (size_t)c * nq + q] : 0.0;
                }
        }
    }
    return BDS_OK;
}

}  // namespace bds

using namespace bds;

extern "C" int bds_track_mem(bds_ctx *ctx, const bds_settings *s, const int8_t *file_bytes, size_t n_bytes, int n_ch,
                             const bds_channel *channel, bds_track_out *out) {
    if (!ctx || !file_bytes) return BDS_ERR_ARG;
    const RecordLoader load = [&](size_t off, size_t n, int8_t *dst) -> int {
        BDS_HIP(ctx, hipMemcpyAsync(dst, file_bytes + off, n, hipMemcpyHostToDevice, (hipStream_t)ctx->stream));
        return BDS_OK;
    };
    return do_track(ctx, s, load, n_bytes, n_ch, channel, out);
}

extern "C" int bds_track(bds_ctx *ctx, const bds_settings *s, const char *path, int n_ch, const bds_channel *channel,
                         bds_track_out *out) {
    if (!ctx || !path) return BDS_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ctx, BDS_ERR_IO, "Unable to read file %s", path);  // postProcessing.m:152-154
    struct Closer {
        FILE *f;
        void *pin = nullptr;
        ~Closer() {
            fclose(f);
            if (pin) (void)hipHostFree(pin);
        }
    } guard{f};
    fseeko(f, 0, SEEK_END);
    const long long sz = ftello(f);
    if (sz <= 0) return fail(ctx, BDS_ERR_IO, "file %s is empty", path);
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    // the window of the record the channels can touch, staged through pinned memory in 256 MiB pieces
    const size_t piece = 256u << 20;
    const RecordLoader load = [&](size_t off, size_t n, int8_t *dst) -> int {
        if (!guard.pin && hipHostMalloc(&guard.pin, std::min<size_t>(piece, (size_t)sz), 0) != hipSuccess)
            return fail(ctx, BDS_ERR_NOMEM, "pinned staging buffer");
        if (fseeko(f, (off_t)off, SEEK_SET)) return fail(ctx, BDS_ERR_IO, "seek in %s failed", path);
        size_t done = 0;
        while (done < n) {
            const size_t m = std::min(piece, n - done);
            if (fread(guard.pin, 1, m, f) != m) return fail(ctx, BDS_ERR_IO, "short read on %s", path);
            const hipError_t e = hipMemcpy(dst + done, guard.pin, m, hipMemcpyHostToDevice);
            if (e != hipSuccess) return fail(ctx, BDS_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
            done += m;
        }
        return BDS_OK;
    };
    return do_track(ctx, s, load, (size_t)sz, n_ch, channel, out);
}

extern "C" long long bds_track_loaded_bytes(bds_ctx *ctx) {
    return (ctx && ctx->trk) ? (long long)ctx->trk->loaded_bytes : 0;
}

extern "C" int bds_track_correlate(bds_ctx *ctx, const bds_settings *s, const int8_t *file_bytes, size_t n_bytes,
                                   int n_ch, const int32_t *prn, const double *state6, double *sums18) {
    if (!ctx || !s || !file_bytes || !prn || !state6 || !sums18 || n_ch < 1) return BDS_ERR_ARG;
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->trk) ctx->trk = new TrackState();
    TrackState &t = *ctx->trk;
    TrkParams p{};
    int rc = fill_params(ctx, *s, p, 1, n_bytes);
    if (rc) return rc;
    p.base = 0, p.win_end = p.n_bytes;  // the whole block is resident (the bounds the debug build checks the loads against)
    if ((rc = ensure_prim(ctx, t, s->signal))) return rc;
    long max_blk = 0;
    for (int c = 0; c < n_ch; ++c) {
        if (prn[c] < 1 || prn[c] > BDS_MAX_PRN) return fail(ctx, BDS_ERR_ARG, "prn[%d] out of range", c);
        max_blk = std::max(max_blk, (long)state6[c * 6 + 1]);
    }
    const int nblocks = (int)std::max<long>(1, (max_blk + p.chunk - 1) / p.chunk);
    int8_t *d_data = nullptr;
    int *d_prn = nullptr;
    double *d_s6 = nullptr, *d_part = nullptr, *d_sums = nullptr;
    struct Scope {  // released on every exit path
        void **p[5];
        ~Scope() {
            for (void **q : p)
                if (*q) (void)hipFree(*q);
        }
    } scope{{(void **)&d_data, (void **)&d_prn, (void **)&d_s6, (void **)&d_part, (void **)&d_sums}};
    BDS_HIP(ctx, hipMalloc((void **)&d_data, n_bytes + kDataSlack));
    BDS_HIP(ctx, hipMalloc((void **)&d_prn, sizeof(int) * n_ch));
    BDS_HIP(ctx, hipMalloc((void **)&d_s6, sizeof(double) * 6 * n_ch));
    BDS_HIP(ctx, hipMalloc((void **)&d_part, sizeof(double) * (size_t)n_ch * nblocks * kNSums));
    BDS_HIP(ctx, hipMalloc((void **)&d_sums, sizeof(double) * (size_t)n_ch * kNSums));
    BDS_HIP(ctx, hipMemcpyAsync(d_data, file_bytes, n_bytes, hipMemcpyHostToDevice, st(ctx)));
    BDS_HIP(ctx, hipMemcpyAsync(d_prn, prn, sizeof(int) * n_ch, hipMemcpyHostToDevice, st(ctx)));
    BDS_HIP(ctx, hipMemcpyAsync(d_s6, state6, sizeof(double) * 6 * n_ch, hipMemcpyHostToDevice, st(ctx)));
    dim3 gc(nblocks, n_ch);
    BDS_TRK_LAUNCH(k_trk_correlate_open, gc, (p.runs ? runs_lds_bytes(p.runs, p.prec) : 0), st(ctx), (const int8_t *)d_data, (const int8_t *)t.d_prim, p,
                   (const int *)d_prn, (const double *)d_s6, d_part, nblocks);
    hipLaunchKernelGGL(k_trk_reduce_open, dim3(n_ch), dim3(64), 0, st(ctx), (const double *)d_part, nblocks, d_sums);
    BDS_HIP(ctx, hipGetLastError());
    BDS_HIP(ctx, hipMemcpyAsync(sums18, d_sums, sizeof(double) * (size_t)n_ch * kNSums, hipMemcpyDeviceToHost, st(ctx)));
    BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));
    return BDS_OK;
}

BDS_DEBUG_TU_READER(bds_debug_failures_track)
