// Acquisition: host orchestration of the PRN x Doppler parallel code-phase search.
//
// Replaces BDS-3_B2a/acquisition.m:126-336 and BDS-3_B1C/acquisition.m:125-307 (and
// B1C/GPU_acquisition.m, which is the same algorithm on gpuArray built-ins).
//
// How the search is organised here (DESIGN.md has the full derivation):
//  * The reference's circular correlation of length N (2 code periods, code zero beyond
//    X samples) is evaluated as a linear correlation of the code with the periodic
//    extension of the wiped-off block, zero padded to a 5-smooth length
//    L >= N + X - 1 -- identical lag for lag, and L has no factor 53 (N = 2^2 3 5^5 53 at
//    99.375 MS/s).
//  * Forward transforms run once per Doppler bin (the reference redoes them per PRN),
//    code spectra once per PRN and are cached in the context.
//  * Per (PRN, bin) cell: one row-pass kernel (spectrum product + inverse rows) and one
//    column-pass kernel (inverse columns + |.| combine + max/argmax).  The D x N
//    results matrix is never materialised.
//  * fp32 is only a sieve: every cell within 2e-5 of a PRN's fp32 maximum (plus its
//    +-1 bin / +-1 lag neighbours) is re-evaluated in f64 by direct time-domain
//    correlation; peakSize, secondPeakSize, the fine-Doppler sums, sigPower and the
//    DC mean are all f64 / exact-integer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <map>
#include <set>
#include <tuple>

#include "bds_acq_scols.h"
#include "bds_acq_wcols.h"
#include "bds_acq_refine.h"
#include "bds_acq_wrows.h"
#include "bds_internal.h"

namespace bds {

static const double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------
struct Plan2D {
    long L = 0;
    int L1 = 0, L2 = 0;
    Plan1D p1{}, p2{};  // p1: columns (length L1), p2: rows (length L2)
    TwiddleL twl{};
    int logT = 0, Spad = 0, nt_cols = 0, nt_rows = 0, ntiles = 0;
    size_t lds_cols = 0, lds_rows = 0;
    bool fast = false;  // both lengths have compile-time specialised search kernels (bds_acq_fast.h)
    bool small = false; // 80 x 4096: wave-private row pass + one-lane-per-column pass (bds_acq_scols.h); fp16 storage, two components
    float2 *d_tw80 = nullptr;  // w80^k of that column pass
    float2 *d_tw1 = nullptr, *d_tw2 = nullptr, *d_hi = nullptr, *d_lo = nullptr;
    float2 *d_ftab1 = nullptr, *d_ftab2 = nullptr;  // fp32 stage-twiddle tables of the inverse column / row transform
    float2 *d_wtab = nullptr;                       // per-lane twiddle table of the wave-private column pass (bds_acq_wcols.h)
    float2 *d_wrtab = nullptr;                      // ... of the wave-private 4096-point row pass (bds_acq_wrows.h)
    unsigned long long *d_clk = nullptr;            // clock probe sums (BDS_ACQ_CLOCKPROBE): rows {shader, reference}, columns {shader, reference}
};

static bool is_5smooth(long v) {
    for (int p : {2, 3, 5})
        while (v % p == 0) v /= p;
    return v == 1;
}

static void factor_radices(int S, Plan1D &p) {
    // few, large stages: 16s, then one 8/4/2 for the remaining power of two, then 5s and 3s.
    // The largest radix goes first: the first autosort stage (Ns = 1) needs no twiddles.
    int v = S, n = 0;
    int rad[kMaxStages];
    while (v % 16 == 0) rad[n++] = 16, v /= 16;
    if (v % 8 == 0) rad[n++] = 8, v /= 8;
    if (v % 4 == 0) rad[n++] = 4, v /= 4;
    if (v % 2 == 0) rad[n++] = 2, v /= 2;
    while (v % 5 == 0) rad[n++] = 5, v /= 5;
    while (v % 3 == 0) rad[n++] = 3, v /= 3;
    std::sort(rad, rad + n, [](int a, int b) { return a > b; });
    p.S = S;
    p.nstage = n;
    int ns = 1;
    for (int i = 0; i < n; ++i) {
        p.radix[i] = rad[i];
        p.nb[i] = FastDiv((uint32_t)(S / rad[i]));
        p.ns[i] = FastDiv((uint32_t)ns);
        p.tws[i] = S / (ns * rad[i]);
        ns *= rad[i];
    }
}

static constexpr int kMaxColLen = 1280;   // column-pass transform length limit (LDS: T*L1*8 B)
static constexpr int kMaxRowLen = 8192;   // row-pass transform length limit
static constexpr int kColPoints = 8192;   // T*L1 budget (<= 72 KiB of LDS: two workgroups per CU)

// Relative cost of one length-S LDS transform per point: every stage is an LDS round trip
// (dominant) plus radix-dependent arithmetic.
static double plan_cost(int S) {
    Plan1D p{};
    factor_radices(S, p);
    if (p.nstage > kMaxStages) return 1e30;
    double c = 0;
    for (int i = 0; i < p.nstage; ++i) {
        switch (p.radix[i]) {
            case 2: c += 1.0; break;
            case 3: c += 1.1; break;
            case 4: c += 1.1; break;
            case 5: c += 1.3; break;
            case 8: c += 1.3; break;
            default: c += 1.6; break;
        }
    }
    return c;
}

static bool fast_cols(int a) { return a == 256 || a == 512 || a == 768 || a == 1024; }
static bool fast_rows(int b) { return b == 1280 || b == 2048 || b == 3072 || b == 4096; }

// Padded length L >= need (5-smooth) and its split L1 x L2, chosen by a cost model:
// L * (stage costs of both passes + a memory term) -- a slightly longer transform made of
// radix-16 stages beats the tightest 5-smooth length made of 3s and 5s.
// small_ok: the 80 x 4096 plan may be chosen (small_plan_ok(): two components, fp16 storage, the specialised kernels on, and
// every searched lag inside the output rows k_cols_small_f forms)
static bool choose_lengths(const Tuning &tune, long need, long &L, int &L1, int &L2, bool small_ok) {
    const double kMem = 3.0;  // HBM/L2 traffic of the two passes, in units of one LDS stage
    double best = 1e30;
    const long lo = std::max<long>(need, 64), hi = lo + lo / 2 + 64;
    if (tune.force_l1 > 0) {  // BDS_ACQ_FORCE_L1L2 (tuning / tests)
        const int a = tune.force_l1, b = tune.force_l2;
        if ((long)a * b >= need && is_5smooth(a) && is_5smooth(b) && a <= kMaxColLen && b <= kMaxRowLen) {
            L = (long)a * b;
            L1 = a;
            L2 = b;
            return true;
        }
    }
    for (long cand = lo; cand <= hi; ++cand) {
        if (!is_5smooth(cand)) continue;
        for (int a = 4; a <= kMaxColLen; ++a) {
            if (cand % a) continue;
            const long b = cand / a;
            if (b > kMaxRowLen || b < a / 4) continue;
            double c = (double)cand * (plan_cost(a) + plan_cost((int)b) + kMem);
            if (fast_cols(a) && fast_rows((int)b)) c *= 0.6;  // specialised kernels exist
            // 80 x 4096 (round 4): 4096-point rows on the wave-private row pass, 80-point columns one lane each -- measured
            // against 256 x 1280 at cfg2: see DESIGN.md 1.6
            if (small_ok && a == kSColsLen && b == 4096) c *= 0.4;
            if (c < best) best = c, L = cand, L1 = a, L2 = (int)b;
        }
    }
    return best < 1e29;
}

// threads a workgroup needs for T transforms of plan p: every stage must fit
// (S/R)*T butterflies into floor(16/R) per thread, and loaders hold <= 16 points per thread
static int threads_for(const Plan1D &p, int T) {
    long need = ((long)p.S * T + kPointsPerThread - 1) / kPointsPerThread;
    for (int i = 0; i < p.nstage; ++i) {
        const int R = p.radix[i], mb = kPointsPerThread / R;
        need = std::max<long>(need, ((long)(p.S / R) * T + mb - 1) / mb);
    }
    need = ((need + 63) / 64) * 64;
    return (int)std::max<long>(64, need);
}

static void plan_free(Plan2D &pl) {
    for (float2 **p : {&pl.d_tw1, &pl.d_tw2, &pl.d_hi, &pl.d_lo, &pl.d_ftab1, &pl.d_ftab2, &pl.d_wtab, &pl.d_wrtab, &pl.d_tw80})
        if (*p) (void)hipFree(*p), *p = nullptr;
    if (pl.d_clk) (void)hipFree(pl.d_clk), pl.d_clk = nullptr;
}

// fp32 stage tables of an inverse transform (bds_fft_t.h tstage TAB): per stage after the first [q][k], entries
// exp(+2 pi j q k / (NS R))
static int upload_stage_tables_f32(bds_ctx *ctx, const Plan1D &p, float2 **dptr) {
    std::vector<float2> h;
    int ns = 1;
    for (int s = 0; s < p.nstage; ++s) {
        const int R = p.radix[s];
        if (ns > 1)
            for (int q = 0; q < R; ++q)
                for (int k = 0; k < ns; ++k) {
                    const double a = 2.0 * kPi * (double)((long)q * k) / (double)((long)ns * R);
                    h.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
                }
        ns *= R;
    }
    if (h.empty()) h.push_back(make_float2(1.f, 0.f));
    BDS_HIP(ctx, hipMalloc((void **)dptr, sizeof(float2) * h.size()));
    BDS_HIP(ctx, hipMemcpy(*dptr, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
    return BDS_OK;
}

// per-lane twiddle table of the wave-private column pass (layout: wcols_table_entries<S>() in bds_acq_wcols.h), inverse
// direction, rounded from f64: [p - 1][thread] = w_S^(b p) with b = 16 (thread / 64) + (thread % 64) / 4, then
// [j - 1][lane] = w_64^(u j) with u = lane / 8 (stage 3 applies the stage-2 twiddle to its inputs, input j being bl = j)
static int upload_wcols_table(bds_ctx *ctx, int S, float2 **dptr) {
    const int R1 = S / 64;
    std::vector<float2> h;
    for (int p = 1; p < R1; ++p)
        for (int t = 0; t < 256; ++t) {
            const int b = 16 * (t >> 6) + ((t & 63) >> 2);
            const double a = 2.0 * kPi * (double)((b * p) % S) / (double)S;
            h.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
        }
    for (int j = 1; j < 8; ++j)
        for (int lane = 0; lane < 64; ++lane) {
            const int u = lane >> 3;
            const double a = 2.0 * kPi * (double)((u * j) % 64) / 64.0;
            h.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
        }
    BDS_HIP(ctx, hipMalloc((void **)dptr, sizeof(float2) * h.size()));
    BDS_HIP(ctx, hipMemcpy(*dptr, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
    return BDS_OK;
}

// per-lane twiddle table of the wave-private 4096-point row pass (layout: bds_acq_wrows.h), inverse direction, rounded from f64
static int upload_wrows_table(bds_ctx *ctx, float2 **dptr) {
    std::vector<float2> h;
    for (int p = 1; p < 16; ++p)
        for (int b = 0; b < 256; ++b) {
            const double a = 2.0 * kPi * (double)((b * p) % 4096) / 4096.0;
            h.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
        }
    for (int j = 1; j < 16; ++j)
        for (int lane = 0; lane < 64; ++lane) {
            const int u = lane >> 2, bl = (j + u) & 15;
            const double a = 2.0 * kPi * (double)(((u * (bl - u)) % 256 + 256) % 256) / 256.0;
            h.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
        }
    for (int k = 0; k < 16; ++k) {
        const double a = 2.0 * kPi * (double)k / 16.0;
        h.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
    }
    for (int u = 0; u < 16; ++u) {
        const double a = 2.0 * kPi * (double)((u * u) % 256) / 256.0;
        h.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
    }
    BDS_HIP(ctx, hipMalloc((void **)dptr, sizeof(float2) * h.size()));
    BDS_HIP(ctx, hipMemcpy(*dptr, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
    return BDS_OK;
}

static int upload_twiddles(bds_ctx *ctx, int n, long denom, long step, float2 **dptr) {
    // table[i] = exp(-2 pi j * (i*step) / denom), computed in f64
    std::vector<float2> h((size_t)n);
    for (int i = 0; i < n; ++i) {
        const long m = ((long)i * step) % denom;
        const double a = -2.0 * kPi * (double)m / (double)denom;
        h[i] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    BDS_HIP(ctx, hipMalloc((void **)dptr, sizeof(float2) * (size_t)n));
    BDS_HIP(ctx, hipMemcpy(*dptr, h.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    return BDS_OK;
}

// The register column pass of the 80 x 4096 plan forms output rows 0 .. kSColsOut - 1 only (bds_acq_scols.h): the plan is
// eligible -- and gets its cost bonus in choose_lengths -- only when the largest searched lag N - 1 lies in those rows and
// the kernels that need it will really run (the same predicate sets pl.small).  cfg2: N = 198 750 -> row 48.  B2a at
// 102 MS/s (N = 204 000 -> row 49) or B1C with pilot at 25 MS/s, cohT 1 (N = 275 000 -> row 67) stay on 256 x 1280.
static bool small_plan_ok(const Tuning &tune, long n_lags, bool allow_small) {
    return allow_small && !tune.generic && n_lags >= 1 && (n_lags - 1) / 4096 < kSColsOut;
}

static int plan_build(bds_ctx *ctx, Plan2D &pl, long need, long n_lags, bool allow_small) {
    plan_free(pl);
    const Tuning &tune = ctx->tune;
    pl.small = false;
    const bool small_ok = small_plan_ok(tune, n_lags, allow_small);
    if (!choose_lengths(tune, need, pl.L, pl.L1, pl.L2, small_ok))
        return fail(ctx, BDS_ERR_UNSUPPORTED, "no two-pass transform plan for length >= %ld", need);
    factor_radices(pl.L1, pl.p1);
    factor_radices(pl.L2, pl.p2);
    int logT = 5;
    while (logT > 0 && ((long)pl.L1 << logT) > kColPoints) --logT;
    while (logT > 0 && (1 << logT) > pl.L2) --logT;
    if (tune.logt >= 0) logT = std::max(0, std::min(logT, tune.logt));  // tuning
    const bool want_fast = fast_cols(pl.L1) && fast_rows(pl.L2) && !tune.generic;
    if (want_fast) {  // the specialised column kernels are built for T = 8 (default; fp16-arithmetic ones also T = 4)
        // 8 columns per workgroup: a tile row is 32 bytes, shared by two lanes (cfg3 search 201.6 -> 196.0 ms,
        // cfg2 3.09 -> 2.64 ms against T = 4, with 768 x 8 on 512 threads; on 384 threads it was 241 ms)
        logT = tune.logt == 2 ? 2 : 3;
    }
    pl.logT = logT;
    pl.Spad = lds_span(pl.L1) + 4;  // +4: successive columns start 8 dwords apart in the bank row
    pl.ntiles = (pl.L2 + (1 << logT) - 1) >> logT;
    pl.nt_cols = threads_for(pl.p1, 1 << logT);
    pl.nt_rows = threads_for(pl.p2, 1);
    if (pl.nt_cols > 1024 || pl.nt_rows > 1024)
        return fail(ctx, BDS_ERR_UNSUPPORTED, "transform %d x %d exceeds the per-workgroup budget", pl.L1, pl.L2);
    pl.lds_cols = sizeof(float2) * (size_t)pl.Spad * (size_t)(1 << logT);
    pl.lds_rows = sizeof(float2) * (size_t)lds_span(pl.L2);
    pl.fast = want_fast;
    pl.small = small_ok && pl.L1 == kSColsLen && pl.L2 == 4096;
    if (tune.verbose) {
        fprintf(stderr, "[bds] search kernels: %s\n", pl.fast ? "specialised" : "generic");
        fprintf(stderr, "[bds] plan: need %ld -> L %ld = %d (cols:", need, pl.L, pl.L1);
        for (int i = 0; i < pl.p1.nstage; ++i) fprintf(stderr, " %d", pl.p1.radix[i]);
        fprintf(stderr, "; T=%d, %d thr, %zu B LDS) x %d (rows:", 1 << logT, pl.nt_cols, pl.lds_cols, pl.L2);
        for (int i = 0; i < pl.p2.nstage; ++i) fprintf(stderr, " %d", pl.p2.radix[i]);
        fprintf(stderr, "; %d thr, %zu B LDS)\n", pl.nt_rows, pl.lds_rows);
    }
    int rc;
    if ((rc = upload_twiddles(ctx, pl.L1, pl.L1, 1, &pl.d_tw1))) return rc;
    if ((rc = upload_twiddles(ctx, pl.L2, pl.L2, 1, &pl.d_tw2))) return rc;
    const int nhi = (int)((pl.L + (1L << kTwLoBits) - 1) >> kTwLoBits);
    if ((rc = upload_twiddles(ctx, nhi, pl.L, 1L << kTwLoBits, &pl.d_hi))) return rc;
    if ((rc = upload_twiddles(ctx, 1 << kTwLoBits, pl.L, 1, &pl.d_lo))) return rc;
    if (pl.fast) {
        if ((rc = upload_stage_tables_f32(ctx, pl.p1, &pl.d_ftab1))) return rc;
        if ((rc = upload_stage_tables_f32(ctx, pl.p2, &pl.d_ftab2))) return rc;
        if ((rc = upload_wcols_table(ctx, pl.L1, &pl.d_wtab))) return rc;
        if (pl.L2 == 4096 && (rc = upload_wrows_table(ctx, &pl.d_wrtab))) return rc;
        BDS_HIP(ctx, hipMalloc((void **)&pl.d_clk, 4 * sizeof(unsigned long long)));
        BDS_HIP(ctx, hipMemset(pl.d_clk, 0, 4 * sizeof(unsigned long long)));
    }
    if (pl.small) {
        if ((rc = upload_wrows_table(ctx, &pl.d_wrtab))) return rc;
        std::vector<float2> h(kSColsLen);
        for (int k = 0; k < kSColsLen; ++k) {
            const double ang = 2.0 * kPi * (double)k / (double)kSColsLen;
            h[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
        BDS_HIP(ctx, hipMalloc((void **)&pl.d_tw80, sizeof(float2) * h.size()));
        BDS_HIP(ctx, hipMemcpy(pl.d_tw80, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
    }
    pl.p1.tw = pl.d_tw1;
    pl.p2.tw = pl.d_tw2;
    pl.twl.hi = pl.d_hi;
    pl.twl.lo = pl.d_lo;
    return BDS_OK;
}

// ---------------------------------------------------------------------------------------
struct PrnResult {
    double peak = 0, denom = 0;
    int fbin = 0;  // 1-based
    long codePhase = 0;
    bool detected = false;
};

struct AcqState {
    // key of everything cached below
    int signal = 0, pilotACQ = 0, code_len = 0;
    double fs = 0, cfb = 0, cohT = 0;
    long spc = 0, X = 0, N = 0, n_ext = 0;
    int ncomp = 0;
    Plan2D plan;
    CodeTable tab{};   // xlen = X (coarse)
    // device buffers
    int8_t *d_sig = nullptr;        // int8 record as loaded (pairs when complex)
    size_t sig_cap = 0;
    double *d_sig64 = nullptr;      // conditioned f64 block of the resampling branch (acquisition.m:56-124)
    size_t sig64_cap = 0;
    double *d_ffa = nullptr, *d_ffb = nullptr, *d_fir = nullptr;  // filtfilt work buffers, fir1 taps
    size_t ffa_cap = 0, ffb_cap = 0, fir_cap = 0;
    int skind = kS8;                // what the search reads: SampleKind
    long n_samples = 0;             // samples of the block the search sees (after resampling, if any)
    ResamplePlan rs;                // resampling branch of the loaded block
    std::vector<double> h_re, h_im;       // host copy of that block (exact for int8 data)
    std::vector<double> h_prefix;         // prefix sums (DC means; integers below 2^53 for int8 data)
    std::vector<double> h_prefix_q;       // ... of the imaginary part
    bool cplx = false;              // longSignal = I + 1i*Q (postProcessing.m:92-96)
    SampleView sview() const { return SampleView{skind >= kF64 ? (const void *)d_sig64 : (const void *)d_sig, skind, n_samples}; }
    int8_t *d_prim = nullptr;       // [63][2][code_len]
    float2 *d_Cs = nullptr;         // [slots][ncomp][L]
    size_t cs_cap_slots = 0;
    std::map<int, int> cs_slot;     // PRN -> slot
    float2 *d_Xs = nullptr;
    size_t xs_cap = 0;  // elements
    float2 *d_Bw = nullptr;
    size_t bw_cap = 0;  // elements
    Rec *d_recs = nullptr;
    size_t recs_cap = 0;
    float *d_rowmax = nullptr;
    int *d_rowarg = nullptr;
    size_t rows_cap = 0;
    int8_t *d_codes = nullptr;       // sampled codes [slot*2 + mode][code_stride] for the f64 sums
    long code_stride = 0;
    std::vector<char> code_have;     // which (slot, mode) tables exist
    char *d_cells = nullptr;         // cell list of the batched second-peak launch
    size_t cells_cap = 0;
    Extra *d_extra = nullptr;        // overflow list of the column pass (bds_acq_f32.h)
    size_t extra_cap = 0;
    int *d_extra_count = nullptr;
    int n_extra_last = 0;            // entries of the last search (diagnostics)
    unsigned long long *d_cellmax = nullptr;  // wave-private column pass: packed {maximum, first lag} per cell ...
    size_t cellmax_cap = 0;
    float *d_lb = nullptr;                    // ... and the running lower bound of each PRN's sieve maximum
    size_t lb_cap = 0;
    std::map<int, std::vector<std::pair<int, long>>> last_cands;  // PRN -> (bin, lag) cells the last run refined in f64
    bool no_fast_search = false;     // this configuration fell back to the run-time-plan search kernels
    bool no_small = false;           // this configuration's 80 x 4096 plan fell back to fp32 storage: re-planned without it
    CorrJob *d_jobs = nullptr;
    double2 *d_jobout = nullptr;
    size_t jobs_cap = 0;
    // device refinement chain (bds_acq_refine.h)
    char *d_ref_zero = nullptr;         // one block, zeroed per run: RefGlobal | RefPrn[P] | cellmax2[P] | lb2[P] | extra2_count
    size_t ref_zero_cap = 0;
    RefPrn *d_ref_prn = nullptr;        // (pointers into d_ref_zero)
    RefGlobal *d_ref_g = nullptr;
    RefCand *d_ref_cand = nullptr;      // [kRefCandCap] coarse candidates, then [kRefCandCap] of the second-peak pass
    size_t ref_cand_cap = 0;
    char *d_ref_tabs = nullptr;         // per run: PRN of index pi (int), code-spectrum offset of index pi (long)
    size_t ref_tabs_cap = 0;
    double *d_prefix_c = nullptr, *d_prefix_cq = nullptr;  // prefix sums of the block at every 256th sample (exact integers: int8 data)
    size_t prefix_c_cap = 0, prefix_cq_cap = 0;
    Extra *d_extra2 = nullptr;          // candidate list / per-PRN maxima / bounds of the B2a second-peak pass
    size_t extra2_cap = 0;
    int *d_extra2_count = nullptr;      // (these three: pointers into d_ref_zero)
    unsigned long long *d_cellmax2 = nullptr;
    float *d_lb2 = nullptr;
    int cands_on_device = 0;            // >0: last_cands of the last run still sits in d_ref_cand (fetched on demand)
    std::vector<int> cands_prns;
    // last run (diagnostics)
    int D = 0;
    std::vector<int> run_prns;
    std::vector<float> h_rowmax;
    std::vector<int> h_rowarg;
    std::map<int, PrnResult> last;
    // (PRN, bin) cells per launch pair; 0 = all Doppler bins of the PRN that fit the work-buffer budget.
    // Measured on the B1C plan (us/cell): 1 -> 41, 4 -> 25, 16 -> 21 without the row-pass cell loop;
    // with it 16 -> 17.5, 48 -> 16.5, 201 -> 15.9 (tools/exp/exp_gchunk.sh)
    int group_env = 0;
    int group = 16;
    bool half = false;         // spectra + inter-pass buffer stored as fp16 complex (specialised plans only)
    double sum_abs_ext = 0;    // sum |x| over the periodically extended block: bound of |X[k]|
    double sum_sq_ext = 0;     // sum x^2 over it: X_rms^2 (Parseval)
    long sums_N = 0, sums_next = 0;  // sizes the two sums above were computed for
    float sX = 1.f, sC = 1.f, sB = 1.f;  // power-of-two storage scales
    long sigpower_X = 0;       // X the cached B1C normaliser was computed for (0: none; reset by bds_acq_load)
    double sigpower = 0;       // sqrt(var(sig(1:X)) * X), B1C/acquisition.m:150
};

void acq_state_free(AcqState *a) {
    if (!a) return;
    plan_free(a->plan);
    for (void *p : {(void *)a->d_sig, (void *)a->d_prim, (void *)a->d_Cs, (void *)a->d_Xs, (void *)a->d_Bw,
                    (void *)a->d_recs, (void *)a->d_rowmax, (void *)a->d_rowarg, (void *)a->d_jobs, (void *)a->d_codes,
                    (void *)a->d_jobout, (void *)a->d_sig64, (void *)a->d_ffa, (void *)a->d_ffb, (void *)a->d_fir, (void *)a->d_cells,
                    (void *)a->d_extra, (void *)a->d_extra_count, (void *)a->d_cellmax, (void *)a->d_lb, (void *)a->d_ref_zero,
                    (void *)a->d_ref_cand, (void *)a->d_ref_tabs, (void *)a->d_prefix_c, (void *)a->d_prefix_cq, (void *)a->d_extra2})
        if (p) (void)hipFree(p);
    delete a;
}

// After the tuning knobs changed: the plan, the storage mode and the element order of the cached spectra all depend on them
// (spectra_permuted() must agree between the forward transforms and the search), so the next bds_acq_prepare re-derives
// everything.  The loaded IF block stays.
void acq_state_invalidate(AcqState *a) {
    if (a) a->plan.L = 0;
}

static int check_settings(bds_ctx *ctx, const bds_settings &s) {
    if (s.signal != BDS_SIGNAL_B1C && s.signal != BDS_SIGNAL_B2A)
        return fail(ctx, BDS_ERR_ARG, "settings.signal must be 1 (B1C) or 2 (B2a)");
    if (s.dataType != 0)
        return fail(ctx, BDS_ERR_UNSUPPORTED, "settings.dataType: only 'schar' (int8 samples) is supported");
    if (!(s.samplingFreq > 0) || !(s.codeFreqBasis > 0) || s.codeLength != 10230)
        return fail(ctx, BDS_ERR_ARG, "settings.samplingFreq/codeFreqBasis must be positive and codeLength 10230");
    if (!(s.acqStep > 0) || !(s.acqSearchBand >= 0))
        return fail(ctx, BDS_ERR_ARG, "settings.acqStep must be > 0 and acqSearchBand >= 0");
    if (s.resamplingflag == 1 && s.samplingFreq > s.resamplingThreshold) {
        const ResamplePlan r = resample_plan(s);
        if (!(r.wp1 > 0 && r.wp2 < 1 && r.wp1 < r.wp2))
            return fail(ctx, BDS_ERR_ARG, "resampling band edges [%g %g] outside (0, 1): fir1 would fail (acquisition.m:66-68)", r.wp1, r.wp2);
    }
    if (s.n_acq < 1 || s.n_acq > BDS_MAX_PRN) return fail(ctx, BDS_ERR_ARG, "settings.acqSatelliteList is empty or too long");
    for (int i = 0; i < s.n_acq; ++i)
        if (s.acqSatelliteList[i] < 1 || s.acqSatelliteList[i] > BDS_MAX_PRN)
            return fail(ctx, BDS_ERR_ARG, "settings.acqSatelliteList[%d] = %d out of 1..63", i, s.acqSatelliteList[i]);
    if (s.signal == BDS_SIGNAL_B1C && !(s.acqCohT > 0 && s.acqCohT <= 10))
        return fail(ctx, BDS_ERR_ARG, "settings.acqCohT must be in (0, 10] ms");
    if (s.signal == BDS_SIGNAL_B2A && s.fineNoncoh < 1)
        return fail(ctx, BDS_ERR_ARG, "settings.fineNoncoh must be >= 1");
    return BDS_OK;
}

template <class T>
static int ensure(bds_ctx *ctx, T **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return BDS_OK;
    if (*p) (void)hipFree(*p), *p = nullptr, *cap = 0;
    hipError_t e = hipMalloc((void **)p, sizeof(T) * need);
    if (e != hipSuccess)
        return fail(ctx, BDS_ERR_NOMEM, "hipMalloc of %zu bytes failed: %s", sizeof(T) * need, hipGetErrorString(e));
    *cap = need;
    return BDS_OK;
}

// (Re)derive sizes, plan and code tables when the settings that define them change.
static int acq_configure(bds_ctx *ctx, const bds_settings &s) {
    int rc = check_settings(ctx, s);
    if (rc) return rc;
    if (!ctx->acq) ctx->acq = new AcqState();
    AcqState &a = *ctx->acq;
    const long spc = samples_per_code(s);
    long X, N;
    int ncomp;
    if (s.signal == BDS_SIGNAL_B1C) {
        X = (long)m_round((double)spc / 10 * s.acqCohT);         // samplesXmsLen  B1C/acquisition.m:132
        N = (long)m_round((double)spc / 10 * (10 + s.acqCohT));  // len10PlusXms   :135
        ncomp = s.pilotACQflag == 1 ? 2 : 1;
    } else {
        X = spc;      // B2a/acquisition.m:179-180
        N = 2 * spc;  // len2ms :134
        ncomp = 2;
    }
    if (X < 1 || X > spc || N <= X) return fail(ctx, BDS_ERR_ARG, "degenerate acquisition sizes (spc=%ld X=%ld N=%ld)", spc, X, N);
    const bool same_key = a.signal == s.signal && a.fs == s.samplingFreq && a.cfb == s.codeFreqBasis &&
                          a.cohT == s.acqCohT && a.pilotACQ == s.pilotACQflag && a.code_len == s.codeLength;
    if (same_key && a.plan.L > 0) return BDS_OK;
    if (!same_key) a.no_small = false;  // (set by the fp32-storage fallback of an 80 x 4096 plan, which re-plans with plan.L = 0)
    a.no_fast_search = false;
    a.code_have.clear();  // sampled-code cache: same key
    a.cs_slot.clear();  // the spectra cache is keyed by everything above: drop it (slot size depends on L)
    if (a.d_Cs) (void)hipFree(a.d_Cs), a.d_Cs = nullptr;
    a.cs_cap_slots = 0;
    a.signal = s.signal;
    a.fs = s.samplingFreq;
    a.cfb = s.codeFreqBasis;
    a.cohT = s.acqCohT;
    a.pilotACQ = s.pilotACQflag;
    a.code_len = s.codeLength;
    a.spc = spc;
    a.X = X;
    a.N = N;
    a.n_ext = N + X - 1;
    a.ncomp = ncomp;
    // (the 80 x 4096 plan: two components, fp16 storage, wave-private row pass available; BDS_ACQ_SMALL=0 keeps 256 x 1280)
    const bool allow_small = ncomp == 2 && ctx->tune.fp16_storage != 0 && ctx->tune.small_plan != 0 && ctx->tune.wcols != 0 && !a.no_small;
    if ((rc = plan_build(ctx, a.plan, a.n_ext, N, allow_small))) return rc;
    a.group_env = ctx->tune.group;
    // default: fp32 search arithmetic on fp16-stored spectra (BDS_ACQ_FP16=0: fp32 storage)
    a.half = (a.plan.fast || a.plan.small) && ctx->tune.fp16_storage != 0;
    // code spectrum: |fft(code)| <= X; stored value conj(C)/L * sC, kept below 2^15
    a.sC = a.half ? (float)std::exp2(std::floor(std::log2(32768.0 * (double)a.plan.L / (double)a.X))) : 1.f;
    // primary codes of every PRN, both components
    const size_t prim_bytes = (size_t)BDS_MAX_PRN * 2 * 10230;
    if (!a.d_prim) BDS_HIP(ctx, hipMalloc((void **)&a.d_prim, prim_bytes));
    std::vector<int8_t> prim(prim_bytes);
    for (int prn = 1; prn <= BDS_MAX_PRN; ++prn)
        for (int c = 0; c < 2; ++c) gen_primary(s.signal, c == 1, prn, &prim[((size_t)(prn - 1) * 2 + c) * 10230]);
    BDS_HIP(ctx, hipMemcpy(a.d_prim, prim.data(), prim.size(), hipMemcpyHostToDevice));
    a.tab.prim = a.d_prim;
    a.tab.ts = 1.0 / s.samplingFreq;                                                     // makeDataTable.m:49
    a.tab.tc = s.signal == BDS_SIGNAL_B1C ? 1.0 / s.codeFreqBasis / 2 : 1.0 / s.codeFreqBasis;  // :50 / makeB2aDataTable.m:47
    a.tab.spc = spc;
    a.tab.xlen = X;
    a.tab.code_len = 10230;
    a.tab.boc = s.signal == BDS_SIGNAL_B1C ? 1 : 0;
    return BDS_OK;
}

static hipStream_t st(bds_ctx *ctx) { return (hipStream_t)ctx->stream; }

// raise a kernel's dynamic-LDS limit once per context (= per device)
template <class K>
static void want_lds(bds_ctx *ctx, K kern, size_t bytes) {
    if (ctx->lds_attr_done.insert((const void *)kern).second)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}


static int set_lds_limits(bds_ctx *ctx) {
    if (!ctx->lds_attr_done.insert((const void *)k_rows_fwd).second) return BDS_OK;
    const int maxlds = 160 * 1024 - 4096;  // leave room for the kernels' small static LDS
    BDS_HIP(ctx, hipFuncSetAttribute((const void *)k_cols_fwd<SignalLoader>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    BDS_HIP(ctx, hipFuncSetAttribute((const void *)k_cols_fwd<CodeLoader>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    BDS_HIP(ctx, hipFuncSetAttribute((const void *)k_rows_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    BDS_HIP(ctx, hipFuncSetAttribute((const void *)k_rows_fwd_st<__half2>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    BDS_HIP(ctx, hipFuncSetAttribute((const void *)k_rows_inv<1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    BDS_HIP(ctx, hipFuncSetAttribute((const void *)k_rows_inv<2>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    BDS_HIP(ctx, hipFuncSetAttribute((const void *)k_cols_inv_max<1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    BDS_HIP(ctx, hipFuncSetAttribute((const void *)k_cols_inv_max<2>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    return BDS_OK;
}

// forward transform of `nb` batches produced by loader `ld` into dst[b*dst_stride]
// forward passes on the specialised stages (plans with pl.fast)
template <int S, class Loader>
static void launch_cols_fwd_t(bds_ctx *ctx, hipStream_t s_, const Plan2D &pl, Loader ld, int nb, float2 *Bw) {
#ifndef BDS_FWD_T
#define BDS_FWD_T 4
#endif
    constexpr int T = BDS_FWD_T;
    const size_t lds = sizeof(float2) * (T * tspan<S>() + lds_span(twiddle_entries<S>()));
    want_lds(ctx, k_cols_fwd_t<S, T, Loader>, lds);
    hipLaunchKernelGGL((k_cols_fwd_t<S, T, Loader>), dim3((pl.L2 + T - 1) / T, nb), dim3(cols_threads<S, T>()), lds, s_,
                       (const float2 *)pl.d_tw1, pl.twl, pl.L2, ld, Bw, pl.L);
}
template <int S, class ST>
static void launch_rows_fwd_t(bds_ctx *ctx, hipStream_t s_, const Plan2D &pl, int nb, const float2 *Bw, ST *dst, long dst_stride,
                              int conj_flag, float scale, int perm) {
    const size_t lds = sizeof(float2) * (tspan<S>() + lds_span(twiddle_entries<S>()));
    want_lds(ctx, k_rows_fwd_t<S, ST>, lds);
    hipLaunchKernelGGL((k_rows_fwd_t<S, ST>), dim3(pl.L1, nb), dim3(rows_threads<S>()), lds, s_, (const float2 *)pl.d_tw2, Bw,
                       pl.L, dst, dst_stride, conj_flag, scale, perm);
}
template <class ST>
static void launch_rows_fwd_any(bds_ctx *ctx, hipStream_t s_, const Plan2D &pl, int nb, const float2 *Bw, ST *dst, long dst_stride,
                                int conj_flag, float scale, int perm) {
    switch (pl.L2) {
        case 1280: launch_rows_fwd_t<1280, ST>(ctx, s_, pl, nb, Bw, dst, dst_stride, conj_flag, scale, 0); break;
        case 2048: launch_rows_fwd_t<2048, ST>(ctx, s_, pl, nb, Bw, dst, dst_stride, conj_flag, scale, 0); break;
        case 3072: launch_rows_fwd_t<3072, ST>(ctx, s_, pl, nb, Bw, dst, dst_stride, conj_flag, scale, 0); break;
        default: launch_rows_fwd_t<4096, ST>(ctx, s_, pl, nb, Bw, dst, dst_stride, conj_flag, scale, perm); break;
    }
}

// The spectra of this configuration are stored in the element order of the wave-private 4096-point row pass (wrows_perm(),
// bds_acq_fast.h): exactly when launch_rows_f will run k_rows_wave_f on them -- fp16 storage, 4096-point rows, a specialised
// plan (or the 80 x 4096 one) and the specialised search not switched off.  Every fallback (fp32 storage, run-time-plan
// kernels) clears a.half and re-runs bds_acq_prepare, which rebuilds the spectra in natural order.
static bool spectra_permuted(const bds_ctx *ctx, const AcqState &a) {
    const Plan2D &pl = a.plan;
    return a.half && pl.L2 == 4096 && (pl.fast || pl.small) && (ctx->tune.wrows != 0 || pl.small) && !a.no_fast_search;
}

template <class Loader>
static int forward(bds_ctx *ctx, AcqState &a, Loader ld, int nb, float2 *dst, long dst_stride, int conj_flag,
                   float scale) {
    Plan2D &pl = a.plan;
    const int perm = spectra_permuted(ctx, a) ? 1 : 0;
    if (pl.fast && !ctx->tune.generic_fwd) {
        switch (pl.L1) {
            case 256: launch_cols_fwd_t<256>(ctx, st(ctx), pl, ld, nb, a.d_Bw); break;
            case 512: launch_cols_fwd_t<512>(ctx, st(ctx), pl, ld, nb, a.d_Bw); break;
            case 768: launch_cols_fwd_t<768>(ctx, st(ctx), pl, ld, nb, a.d_Bw); break;
            default: launch_cols_fwd_t<1024>(ctx, st(ctx), pl, ld, nb, a.d_Bw); break;
        }
        if (a.half)
            launch_rows_fwd_any<__half2>(ctx, st(ctx), pl, nb, (const float2 *)a.d_Bw, (__half2 *)dst, dst_stride, conj_flag, scale, perm);
        else
            launch_rows_fwd_any<float2>(ctx, st(ctx), pl, nb, (const float2 *)a.d_Bw, dst, dst_stride, conj_flag, scale, 0);
        BDS_HIP(ctx, hipGetLastError());
        return BDS_OK;
    }
    dim3 g1(pl.ntiles, nb), g2(pl.L1, nb);
    hipLaunchKernelGGL(k_cols_fwd<Loader>, g1, dim3(pl.nt_cols), pl.lds_cols, st(ctx), pl.p1, pl.twl, pl.L2,
                       pl.logT, pl.Spad, ld, a.d_Bw, pl.L);
    if (a.half)  // dst counts in stored elements (fp16 complex)
        hipLaunchKernelGGL(k_rows_fwd_st<__half2>, g2, dim3(pl.nt_rows), pl.lds_rows, st(ctx), pl.p2,
                           (const float2 *)a.d_Bw, pl.L, (__half2 *)dst, dst_stride, conj_flag, scale, perm);
    else
        hipLaunchKernelGGL(k_rows_fwd, g2, dim3(pl.nt_rows), pl.lds_rows, st(ctx), pl.p2, (const float2 *)a.d_Bw,
                           pl.L, dst, dst_stride, conj_flag, scale);
    BDS_HIP(ctx, hipGetLastError());
    return BDS_OK;
}

// optional per-cell descriptors (device arrays) for a launch whose cells are not "one PRN, consecutive bins"
struct CellList {
    const int *bin = nullptr;    // Doppler bin of cell g
    const long *cs = nullptr;    // element offset of its code spectra from the Cs base
    const int4 *rng = nullptr;   // searched lag ranges (lo1, hi1, lo2, hi2)
    int gc = 1;                  // consecutive listed cells that share their code spectra (one row workgroup walks them)
};
// where a column pass reports: per-tile records + the overflow list of the sieve
struct SieveOut {
    Rec *recs = nullptr;
    Extra *extra = nullptr;
    int *extra_count = nullptr;
    int extra_cap = 0;
    int cell0 = 0;     // run-wide index of cell 0 of the launch
    float keep = 1.f;  // 1 - sieve tolerance
    // wave-private column pass (bds_acq_wcols.h) reports per cell / per PRN instead of per tile
    unsigned long long *cellmax = nullptr;
    float *lb = nullptr;
    int lb_div = 1;
    hipEvent_t mid = nullptr;  // recorded between the two passes of a sampled launch pair (timing)
    // overlapped passes: the column pass runs on its own stream behind ev_rows and signals ev_cols
    hipStream_t cols_stream = nullptr;
    hipEvent_t ev_rows = nullptr, ev_cols = nullptr;
};

// ---- fp32-arithmetic search kernels (bds_acq_f32.h): dispatch on the compile-time lengths --------------
template <int S, int NC, class ST>
static void launch_rows_f(bds_ctx *ctx, hipStream_t sr, const Plan2D &pl, const void *Xs, int G, int bin0, const void *Cs,
                          void *Bw, float out_scale, const CellList &cl, bool ilv) {
    const size_t lds = sizeof(float2) * (tspan<S>() + f32_tw_span<S, kF32TabRows>());
    want_lds(ctx, k_rows_inv_f<S, NC, ST>, lds);
    // balanced chunks of at most tune.gchunk cells
    int nch = (G + ctx->tune.gchunk - 1) / ctx->tune.gchunk;
    int gc = (G + nch - 1) / nch;
    if (cl.bin) gc = cl.gc, nch = (G + cl.gc - 1) / cl.gc;  // a workgroup stays inside one PRN's cells
    const int nvb = pl.L1 * nch;  // L1 % 8 == 0 on every specialised plan: virtual workgroup vb sits on XCD vb % 8
    const RowsFArgs A{(const float2 *)(kF32TabRows ? pl.d_ftab2 : pl.d_tw2), pl.twl, Xs, pl.L, pl.L1, G, bin0, Cs, Bw, out_scale, gc, nch, cl.bin, cl.cs, nvb, ctx->tune.clockprobe ? pl.d_clk : nullptr, ilv ? 1 : 0};
    const int grid = ctx->tune.rows_grid > 0 ? std::min(nvb, (ctx->tune.rows_grid + 7) / 8 * 8) : nvb;
    if constexpr (S == 4096 && std::is_same<ST, __half2>::value) {
        if (ctx->tune.wrows != 0 || pl.small) {  // wave-private row pass (bds_acq_wrows.h): per-lane twiddle constants, 4 barriers per cell
            RowsFArgs B = A;
            B.tw = pl.d_wrtab;
            const bool pk = ctx->tune.pk != 0;  // packed-fp32 butterflies (bds_fft_pk.h)
            auto go = [&](auto kern) {
                want_lds(ctx, kern, kWRowsLdsBytes);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kWRowsLdsBytes, sr, B);
            };
            if constexpr (NC == 2) {
                if (ilv) {
                    if (pk) go(k_rows_wave_f<NC, true, true>);
                    else go(k_rows_wave_f<NC, true, false>);
                    return;
                }
            }
            if (pk) go(k_rows_wave_f<NC, false, true>);
            else go(k_rows_wave_f<NC, false, false>);
            return;
        }
    }
    hipLaunchKernelGGL((k_rows_inv_f<S, NC, ST>), dim3(grid), dim3(rows_threads<S>()), lds, sr, A);
}
template <int S, int T, int NC, class ST>
static void launch_cols_ft(bds_ctx *ctx, hipStream_t sc, const Plan2D &pl, int G, const void *Bw, float w0, float w1, int lo1,
                           int hi1, int lo2, int hi2, const SieveOut &so, const CellList &cl) {
    const size_t lds = sizeof(float2) * (T * tspan<S>() + f32_tw_span<S, f32_tab_cols<S>()>());
    const ColsFArgs A{(const float2 *)(f32_tab_cols<S>() ? pl.d_ftab1 : pl.d_tw1), pl.L2, Bw, pl.L, w0, w1, lo1, hi1, lo2, hi2, so.recs, pl.ntiles, cl.rng,
                      so.extra, so.extra_count, so.extra_cap, so.cell0, so.keep};
    const bool masked = cl.rng || !(lo1 == 0 && lo2 > hi2);  // anything but "one range starting at lag 0"
    if (masked) {
        want_lds(ctx, k_cols_inv_max_f<S, T, NC, true, ST>, lds);
        hipLaunchKernelGGL((k_cols_inv_max_f<S, T, NC, true, ST>), dim3(pl.ntiles, G), dim3(cols_threads<S, T>()), lds, sc, A);
    } else {
        want_lds(ctx, k_cols_inv_max_f<S, T, NC, false, ST>, lds);
        hipLaunchKernelGGL((k_cols_inv_max_f<S, T, NC, false, ST>), dim3(pl.ntiles, G), dim3(cols_threads<S, T>()), lds, sc, A);
    }
}
// wave-private column pass: one tile per workgroup, workgroups started by the hardware in list order (see the kernel's
// note on item order)
template <int S, int NC, bool MASKED, class ST, int NV, bool ILV = false, bool PK = false>
static void launch_cols_wm(bds_ctx *ctx, hipStream_t sc, const Plan2D &pl, const WColsArgs &A) {
    using W = WCols<S>;
    want_lds(ctx, k_cols_wave_f<S, NC, MASKED, ST, NV, ILV, PK>, W::kLdsBytes);
    WColsArgs B = A;
    const int quads = A.ntiles / 32;  // per XCD and cell
    B.qchunk = std::max(1, std::min(ctx->tune.wcols_qchunk, quads));
    while (quads % B.qchunk) --B.qchunk;
    hipLaunchKernelGGL((k_cols_wave_f<S, NC, MASKED, ST, NV, ILV, PK>), dim3(A.n_items), dim3(W::NT), W::kLdsBytes, sc, B);
}
template <int S, int NC, class ST>
static void launch_cols_w(bds_ctx *ctx, hipStream_t sc, const Plan2D &pl, int G, const void *Bw, float w0, float w1, int lo1,
                          int hi1, int lo2, int hi2, const SieveOut &so, const CellList &cl, bool ilv) {
    const int ntiles = pl.L2 / WCols<S>::T;  // L2 % 256 == 0 on every specialised plan: ntiles % 32 == 0 (8 XCDs x quads)
    const WColsArgs A{(const float2 *)pl.d_wtab, pl.L2, ntiles, G, G * ntiles, Bw, pl.L, w0, w1, lo1, hi1, lo2, hi2, cl.rng,
                      so.cellmax, so.lb, so.lb_div, so.extra, so.extra_count, so.extra_cap, so.cell0, so.keep, 1,
                      ctx->tune.clockprobe ? pl.d_clk : nullptr};
    const bool masked = cl.rng || !(lo1 == 0 && lo2 > hi2);  // anything but "one range starting at lag 0"
    if constexpr (S == 768 && NC == 2 && std::is_same<ST, __half2>::value) {  // (the plan with 4096-point rows: cfg3)
        if (ilv) {  // k_rows_wave_f<2, true> laid the buffer out [cell][element][component]
            const bool pk = ctx->tune.pk != 0 && ctx->tune.pk != 2;  // packed-fp32 butterflies (BDS_ACQ_PK=2: row pass only)
            if (hi1 / pl.L2 < 6 * 8 * WCols<S>::R1) {
                if (pk) launch_cols_wm<S, NC, false, ST, 6, true, true>(ctx, sc, pl, A);
                else launch_cols_wm<S, NC, false, ST, 6, true>(ctx, sc, pl, A);
            } else {
                if (pk) launch_cols_wm<S, NC, false, ST, 8, true, true>(ctx, sc, pl, A);
                else launch_cols_wm<S, NC, false, ST, 8, true>(ctx, sc, pl, A);
            }
            return;
        }
    }
    if (masked)
        launch_cols_wm<S, NC, true, ST, 8>(ctx, sc, pl, A);
    else if (hi1 / pl.L2 < 6 * 8 * WCols<S>::R1)  // no searched lag beyond output row 48 R1: outputs v = 6, 7 of the last stage unused
        launch_cols_wm<S, NC, false, ST, 6>(ctx, sc, pl, A);
    else
        launch_cols_wm<S, NC, false, ST, 8>(ctx, sc, pl, A);
}
template <int S, int NC, class ST>
static void launch_cols_f(bds_ctx *ctx, hipStream_t sc, const Plan2D &pl, int G, const void *Bw, float w0, float w1, int lo1,
                          int hi1, int lo2, int hi2, const SieveOut &so, const CellList &cl, bool ilv) {
    if (so.cellmax) return launch_cols_w<S, NC, ST>(ctx, sc, pl, G, Bw, w0, w1, lo1, hi1, lo2, hi2, so, cl, ilv);
    if (pl.logT == 2)
        launch_cols_ft<S, 4, NC, ST>(ctx, sc, pl, G, Bw, w0, w1, lo1, hi1, lo2, hi2, so, cl);
    else
        launch_cols_ft<S, 8, NC, ST>(ctx, sc, pl, G, Bw, w0, w1, lo1, hi1, lo2, hi2, so, cl);
}
// one group of cells: row pass, then column pass, on one stream
template <int NC, class ST>
static void launch_fast_f(bds_ctx *ctx, hipStream_t st_, const Plan2D &pl, const void *Xs, int G, int bin0, const void *Cs,
                          void *Bw, float out_scale, float w0, float w1, int lo1, int hi1, int lo2, int hi2, const SieveOut &so,
                          const CellList &cl = {}) {
    // both components of an element side by side in the inter-pass buffer: the wave-private pair of the 768 x 4096 plan (cfg3),
    // unmasked search (one lag range from 0), fp16 storage, two components
    if constexpr (NC == 2 && std::is_same<ST, __half2>::value) {
        if (pl.small) {  // 80 x 4096: wave-private row pass (components interleaved) + one lane per column and component
            launch_rows_f<4096, NC, ST>(ctx, st_, pl, Xs, G, bin0, Cs, Bw, out_scale, cl, true);
            if (so.mid) (void)hipEventRecord(so.mid, st_);
            const SColsArgs A{(const float2 *)pl.d_tw80, pl.L2, G, Bw, pl.L, w0, w1, lo1, hi1, lo2, hi2, cl.rng, so.cellmax, so.lb, so.lb_div,
                              so.extra, so.extra_count, so.extra_cap, so.cell0, so.keep};
            want_lds(ctx, k_cols_small_f<NC>, kSColsLdsBytes);
            hipLaunchKernelGGL((k_cols_small_f<NC>), dim3((unsigned)(G * (pl.L2 / (kSColsNT / 2)))), dim3(kSColsNT), kSColsLdsBytes, st_, A);
            return;
        }
    }
    const bool ilv = NC == 2 && std::is_same<ST, __half2>::value && pl.L1 == 768 && pl.L2 == 4096 && ctx->tune.wrows != 0 &&
                     so.cellmax && ctx->tune.ilv != 0 && !cl.rng && lo1 == 0 && lo2 > hi2;
    switch (pl.L2) {
        case 1280: launch_rows_f<1280, NC, ST>(ctx, st_, pl, Xs, G, bin0, Cs, Bw, out_scale, cl, false); break;
        case 2048: launch_rows_f<2048, NC, ST>(ctx, st_, pl, Xs, G, bin0, Cs, Bw, out_scale, cl, false); break;
        case 3072: launch_rows_f<3072, NC, ST>(ctx, st_, pl, Xs, G, bin0, Cs, Bw, out_scale, cl, false); break;
        default: launch_rows_f<4096, NC, ST>(ctx, st_, pl, Xs, G, bin0, Cs, Bw, out_scale, cl, ilv); break;
    }
    if (so.mid) (void)hipEventRecord(so.mid, st_);
    hipStream_t sc = st_;
    if (so.cols_stream) {
        sc = so.cols_stream;
        (void)hipEventRecord(so.ev_rows, st_);
        (void)hipStreamWaitEvent(sc, so.ev_rows, 0);
    }
    switch (pl.L1) {
        case 256: launch_cols_f<256, NC, ST>(ctx, sc, pl, G, Bw, w0, w1, lo1, hi1, lo2, hi2, so, cl, ilv); break;
        case 512: launch_cols_f<512, NC, ST>(ctx, sc, pl, G, Bw, w0, w1, lo1, hi1, lo2, hi2, so, cl, ilv); break;
        case 768: launch_cols_f<768, NC, ST>(ctx, sc, pl, G, Bw, w0, w1, lo1, hi1, lo2, hi2, so, cl, ilv); break;
        default: launch_cols_f<1024, NC, ST>(ctx, sc, pl, G, Bw, w0, w1, lo1, hi1, lo2, hi2, so, cl, ilv); break;
    }
    if (so.cols_stream) (void)hipEventRecord(so.ev_cols, sc);
}

// inter-pass work buffer: two halves of `group` cells each (float2-sized elements)
static size_t bw_batches(const AcqState &a) { return (size_t)std::max(2 * a.group * a.ncomp, 8); }

// cells per launch pair for a search over D bins: the whole Doppler row of a PRN when the two
// halves of the inter-pass buffer stay under 24 GiB (B1C cfg3: 201 cells, 20 GiB of the 288)
static void pick_group(AcqState &a, const bds_settings &s) {
    const int D = (int)m_round(s.acqSearchBand * 2 / s.acqStep) + 1;
    const double per_cell = 2.0 * a.ncomp * (double)a.plan.L * sizeof(float2);
    const int cap = (int)std::max(1.0, std::floor(24.0 * 1073741824.0 / per_cell));
    a.group = a.group_env ? a.group_env : std::min(cap, 256);
    a.group = std::max(1, std::min(a.group, D));
}

}  // namespace bds

using namespace bds;

// =======================================================================================
// The settings acquisition() works with after its resampling branch reassigned samplingFreq and IF
// (acquisition.m:103,119); everything downstream -- code tables, sizes, frequency bins -- uses these.
static const bds_settings *effective(const bds_settings *s, bds_settings *tmp) {
    const ResamplePlan r = resample_plan(*s);
    if (!r.on) return s;
    *tmp = *s;
    tmp->samplingFreq = r.new_fs;
    tmp->IF = r.new_if;
    tmp->resamplingflag = 0;
    return tmp;
}

// filtfilt(fir1(700, wp), 1, longSignal) + index decimation on the device (acquisition.m:56-112);
// leaves the conditioned block in a.d_sig64 and its host copy in h_re / h_im.
template <int NCH>
static int condition_block(bds_ctx *ctx, AcqState &a, const ResamplePlan &r, long n_in, long *n_out) {
    constexpr int kTaps = 701, kFact = 3 * (kTaps - 1);  // filtfilt: nfact = 3*(nfilt-1)
    if (n_in <= kFact) return fail(ctx, BDS_ERR_ARG, "longSignal (%ld samples) is too short for filtfilt (needs > %d)", n_in, kFact);
    const std::vector<double> b = fir1_bandpass(kTaps, r.wp1, r.wp2);
    const long len = n_in + 2L * kFact;
    int rc;
    if ((rc = ensure(ctx, &a.d_fir, &a.fir_cap, (size_t)kTaps))) return rc;
    if ((rc = ensure(ctx, &a.d_ffa, &a.ffa_cap, (size_t)len * NCH))) return rc;
    if ((rc = ensure(ctx, &a.d_ffb, &a.ffb_cap, (size_t)len * NCH))) return rc;
    BDS_HIP(ctx, hipMemcpyAsync(a.d_fir, b.data(), sizeof(double) * kTaps, hipMemcpyHostToDevice, st(ctx)));
    const dim3 grid(2048), blk(256);
    hipLaunchKernelGGL(k_ff_extend<NCH>, grid, blk, 0, st(ctx), (const int8_t *)a.d_sig, n_in, kFact, a.d_ffa);
    hipLaunchKernelGGL(k_ff_fir<NCH>, grid, blk, sizeof(double) * kTaps, st(ctx), (const double *)a.d_ffa, len,
                       (const double *)a.d_fir, kTaps, 0, a.d_ffb);
    hipLaunchKernelGGL(k_ff_fir<NCH>, grid, blk, sizeof(double) * kTaps, st(ctx), (const double *)a.d_ffb, len,
                       (const double *)a.d_fir, kTaps, 1, a.d_ffa);
    const long sig_len = (long)std::floor((double)(n_in - 1) / r.old_fs * r.new_fs);  // :107
    if (sig_len < 1) return fail(ctx, BDS_ERR_ARG, "resampled longSignal is empty");
    if ((rc = ensure(ctx, &a.d_sig64, &a.sig64_cap, (size_t)sig_len * NCH))) return rc;
    hipLaunchKernelGGL(k_ff_decimate<NCH>, grid, blk, 0, st(ctx), (const double *)a.d_ffa, kFact, sig_len, r.new_fs,
                       r.old_fs, a.d_sig64);
    BDS_HIP(ctx, hipGetLastError());
    std::vector<double> h((size_t)sig_len * NCH);
    BDS_HIP(ctx, hipMemcpyAsync(h.data(), a.d_sig64, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st(ctx)));
    BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));
    a.h_re.resize((size_t)sig_len);
    a.h_im.assign(NCH == 2 ? (size_t)sig_len : 0, 0.0);
    for (long i = 0; i < sig_len; ++i) {
        a.h_re[(size_t)i] = h[(size_t)i * NCH];
        if (NCH == 2) a.h_im[(size_t)i] = h[(size_t)i * NCH + 1];
    }
    *n_out = sig_len;
    return BDS_OK;
}

// sum |x| and sum x^2 over the periodically extended block the search transforms (they set the fp16
// storage scales); keyed by the sizes they were computed for, so a run whose settings changed N re-derives them
static void ext_sums(AcqState &a) {
    a.sum_abs_ext = a.sum_sq_ext = 0;
    for (long i = 0; i < a.n_ext; ++i) {
        const long m = i < a.N ? i : i - a.N;
        const double v = a.cplx ? std::hypot(a.h_re[(size_t)m], a.h_im[(size_t)m]) : std::fabs(a.h_re[(size_t)m]);
        a.sum_abs_ext += v;
        a.sum_sq_ext += v * v;
    }
    a.sums_N = a.N;
    a.sums_next = a.n_ext;
}

extern "C" int bds_acq_load(bds_ctx *ctx, const bds_settings *s_in, const int8_t *samples, size_t n_samples,
                            int is_complex) {
    if (!ctx || !s_in || !samples) return BDS_ERR_ARG;
    if (int rc0 = check_settings(ctx, *s_in)) return rc0;  // (the resampling band edges are only visible here)
    bds_settings eff;
    const bds_settings *s = effective(s_in, &eff);
    // n_samples counts complex samples when is_complex: `samples` then holds 2*n_samples int8 (I,Q pairs)
    const bool cplx = is_complex != 0;
    int rc = acq_configure(ctx, *s);
    if (rc) return rc;
    AcqState &a = *ctx->acq;
    const ResamplePlan r = resample_plan(*s_in);
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nb = n_samples * (cplx ? 2 : 1);
    if ((rc = ensure(ctx, &a.d_sig, &a.sig_cap, nb))) return rc;
    BDS_HIP(ctx, hipMemcpyAsync(a.d_sig, samples, nb, hipMemcpyHostToDevice, st(ctx)));
    a.cplx = cplx;
    a.rs = r;
    long n_eff = (long)n_samples;
    if (r.on) {
        rc = cplx ? condition_block<2>(ctx, a, r, (long)n_samples, &n_eff) : condition_block<1>(ctx, a, r, (long)n_samples, &n_eff);
        if (rc) return rc;
        a.skind = cplx ? kF64C : kF64;
    } else {
        a.skind = cplx ? kS8C : kS8;
        a.h_re.resize(n_samples);
        a.h_im.assign(cplx ? n_samples : 0, 0.0);
        for (size_t i = 0; i < n_samples; ++i) {
            a.h_re[i] = (double)samples[cplx ? 2 * i : i];
            if (cplx) a.h_im[i] = (double)samples[2 * i + 1];
        }
    }
    if (n_eff < a.N)
        return fail(ctx, BDS_ERR_ARG, "longSignal has %ld samples%s; acquisition needs at least %ld (acquisition.m:140)",
                    n_eff, r.on ? " after resampling" : "", a.N);
    a.h_prefix.resize((size_t)n_eff + 1);
    a.h_prefix[0] = 0;
    a.h_prefix_q.clear();
    for (long i = 0; i < n_eff; ++i) a.h_prefix[(size_t)i + 1] = a.h_prefix[(size_t)i] + a.h_re[(size_t)i];
    if (cplx) {
        a.h_prefix_q.resize((size_t)n_eff + 1);
        a.h_prefix_q[0] = 0;
        for (long i = 0; i < n_eff; ++i) a.h_prefix_q[(size_t)i + 1] = a.h_prefix_q[(size_t)i] + a.h_im[(size_t)i];
    }
    a.n_samples = n_eff;
    a.sigpower_X = 0;
    if (!r.on) {
        // every 256th prefix sum for the device refinement chain (the DC of the B1C fine-search block, bds_acq_refine.h)
        const size_t nc = (size_t)(n_eff >> 8) + 1;
        std::vector<double> pc(nc);
        for (size_t i = 0; i < nc; ++i) pc[i] = a.h_prefix[i << 8];
        if ((rc = ensure(ctx, &a.d_prefix_c, &a.prefix_c_cap, nc))) return rc;
        BDS_HIP(ctx, hipMemcpyAsync(a.d_prefix_c, pc.data(), sizeof(double) * nc, hipMemcpyHostToDevice, st(ctx)));
        if (cplx) {
            std::vector<double> pq(nc);
            for (size_t i = 0; i < nc; ++i) pq[i] = a.h_prefix_q[i << 8];
            if ((rc = ensure(ctx, &a.d_prefix_cq, &a.prefix_cq_cap, nc))) return rc;
            BDS_HIP(ctx, hipMemcpyAsync(a.d_prefix_cq, pq.data(), sizeof(double) * nc, hipMemcpyHostToDevice, st(ctx)));
            BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));  // (pq leaves scope)
        }
        BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));      // (pc leaves scope)
    }
    ext_sums(a);
    BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));
    return BDS_OK;
}

extern "C" int bds_acq_prepare(bds_ctx *ctx, const bds_settings *s_in) {
    if (!ctx || !s_in) return BDS_ERR_ARG;
    if (int rc0 = check_settings(ctx, *s_in)) return rc0;  // (the resampling band edges are only visible here)
    bds_settings eff;
    const bds_settings *s = effective(s_in, &eff);
    int rc = acq_configure(ctx, *s);
    if (rc) return rc;
    AcqState &a = *ctx->acq;
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = set_lds_limits(ctx))) return rc;
    Plan2D &pl = a.plan;
    pick_group(a, *s);
    if ((rc = ensure(ctx, &a.d_Bw, &a.bw_cap, bw_batches(a) * (size_t)pl.L))) return rc;
    std::vector<int> todo;
    for (int i = 0; i < s->n_acq; ++i)
        if (!a.cs_slot.count(s->acqSatelliteList[i]) &&
            std::find(todo.begin(), todo.end(), s->acqSatelliteList[i]) == todo.end())
            todo.push_back(s->acqSatelliteList[i]);
    if (todo.empty()) return BDS_OK;
    const size_t need_slots = a.cs_slot.size() + todo.size();
    if (need_slots > a.cs_cap_slots) {
        // grow: spectra are cheap to rebuild, so drop the cache instead of copying
        for (auto &kv : a.cs_slot)
            if (std::find(todo.begin(), todo.end(), kv.first) == todo.end()) todo.push_back(kv.first);
        a.cs_slot.clear();
        if (a.d_Cs) (void)hipFree(a.d_Cs), a.d_Cs = nullptr;
        hipError_t e = hipMalloc((void **)&a.d_Cs, sizeof(float2) * need_slots * a.ncomp * (size_t)pl.L);
        if (e != hipSuccess) return fail(ctx, BDS_ERR_NOMEM, "code-spectrum cache (%zu PRNs) does not fit: %s", need_slots, hipGetErrorString(e));
        a.cs_cap_slots = need_slots;
    }
    // conj(fft([table zeros]))/L per PRN and component (B2a/acquisition.m:175-184, B1C/acquisition.m:174-187)
    const int chunk = (int)bw_batches(a);
    for (int prn : todo) {
        const int slot = (int)a.cs_slot.size();
        CodeLoader ld{a.tab, (prn - 1) * 2};
        // components of one PRN are adjacent code slots: batch index = component
        (void)chunk;
        // half storage: the same byte buffer holds 4-byte elements, so offsets count in those
        float2 *cs_dst = a.half ? (float2 *)((__half2 *)a.d_Cs + (size_t)slot * a.ncomp * pl.L)
                                : a.d_Cs + (size_t)slot * a.ncomp * pl.L;
        if ((rc = forward(ctx, a, ld, a.ncomp, cs_dst, pl.L, 1, (float)((double)a.sC / (double)pl.L)))) return rc;
        a.cs_slot[prn] = slot;
    }
    BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));
    return BDS_OK;
}

namespace bds {

struct Cell {
    int b;     // 0-based bin
    long lag;  // 0-based
    bool operator<(const Cell &o) const { return std::tie(b, lag) < std::tie(o.b, o.lag); }
};

// sampled code of (slot, mode) for the f64 sums: built once, cached in the context (a.d_codes allocated by the caller)
static void make_code_table(bds_ctx *ctx, AcqState &a, int slot, int mode) {
    const size_t t = (size_t)slot * 2 + mode;
    if (a.code_have[t]) return;
    CodeTable full = a.tab;
    full.xlen = a.spc;  // whole table; the coarse jobs read its first X samples
    const long len = mode ? a.code_stride : a.spc;
    hipLaunchKernelGGL(k_make_code, dim3(256), dim3(256), 0, st(ctx), full, slot, mode, len, a.d_codes + t * (size_t)a.code_stride);
    a.code_have[t] = 1;
}

static int ensure_code_cache(bds_ctx *ctx, AcqState &a, const bds_settings &s) {
    const long stride = a.signal == BDS_SIGNAL_B2A ? std::max<long>(a.spc, (long)s.fineNoncoh * a.spc) : a.spc;
    const size_t ntab = (size_t)BDS_MAX_PRN * 2 * 2;
    if (!a.d_codes || a.code_stride != stride || a.code_have.size() != ntab) {
        if (a.d_codes) (void)hipFree(a.d_codes), a.d_codes = nullptr;
        hipError_t e = hipMalloc((void **)&a.d_codes, ntab * (size_t)stride);
        if (e != hipSuccess) return fail(ctx, BDS_ERR_NOMEM, "sampled-code cache: %s", hipGetErrorString(e));
        a.code_stride = stride;
        a.code_have.assign(ntab, 0);
    }
    return BDS_OK;
}

constexpr int kCorrSlices = 8;  // partial sums per job (k_corr_f64 grid.y)

static int ensure_job_buffers(bds_ctx *ctx, AcqState &a, size_t njobs) {
    if (a.jobs_cap >= njobs) return BDS_OK;
    int rc;
    if (a.d_jobs) (void)hipFree(a.d_jobs), a.d_jobs = nullptr;
    if (a.d_jobout) (void)hipFree(a.d_jobout), a.d_jobout = nullptr;
    size_t cap = std::max<size_t>(njobs, 1024), dummy = 0;
    if ((rc = ensure(ctx, &a.d_jobs, &dummy, cap))) return rc;
    dummy = 0;
    if ((rc = ensure(ctx, &a.d_jobout, &dummy, cap * kCorrSlices * kCorrFreqs))) return rc;
    a.jobs_cap = cap;
    return BDS_OK;
}

// multi: every job carries up to kCorrFreqs frequencies (k_corr_f64_multi); out[j * kCorrFreqs + f]
static int run_jobs(bds_ctx *ctx, AcqState &a, const bds_settings &s, std::vector<CorrJob> &jobs,
                    std::vector<double2> &out, bool multi = false) {
    const int nper = multi ? kCorrFreqs : 1;
    out.resize(jobs.size() * nper);
    if (jobs.empty()) return BDS_OK;
    int rc;
    // sampled codes the jobs refer to (built once per (slot, mode), cached in the context)
    if ((rc = ensure_code_cache(ctx, a, s))) return rc;
    for (const CorrJob &j : jobs) make_code_table(ctx, a, j.slot, j.mode);
    constexpr int kSlices = kCorrSlices;
    if ((rc = ensure_job_buffers(ctx, a, jobs.size()))) return rc;
    BDS_HIP(ctx, hipMemcpyAsync(a.d_jobs, jobs.data(), sizeof(CorrJob) * jobs.size(), hipMemcpyHostToDevice, st(ctx)));
    if (multi)
        hipLaunchKernelGGL(k_corr_f64_multi, dim3((unsigned)jobs.size(), kSlices), dim3(256), 0, st(ctx), a.sview(), a.N,
                           (const int8_t *)a.d_codes, a.code_stride, 1.0 / a.fs, (const CorrJob *)a.d_jobs, a.d_jobout);
    else
        hipLaunchKernelGGL(k_corr_f64, dim3((unsigned)jobs.size(), kSlices), dim3(256), 0, st(ctx), a.sview(), a.N,
                           (const int8_t *)a.d_codes, a.code_stride, 1.0 / a.fs, (const CorrJob *)a.d_jobs, a.d_jobout, (const int *)nullptr, 0, 0);
    BDS_HIP(ctx, hipGetLastError());
    std::vector<double2> part(jobs.size() * kSlices * nper);
    BDS_HIP(ctx, hipMemcpyAsync(part.data(), a.d_jobout, sizeof(double2) * part.size(), hipMemcpyDeviceToHost, st(ctx)));
    BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));
    for (size_t j = 0; j < jobs.size(); ++j)
        for (int f = 0; f < (multi ? jobs[j].nf : 1); ++f) {
            double2 acc = make_double2(0.0, 0.0);  // slices in order, as a single-frequency job adds them
            for (int k = 0; k < kSlices; ++k) acc.x += part[(j * kSlices + k) * nper + f].x, acc.y += part[(j * kSlices + k) * nper + f].y;
            out[j * nper + f] = acc;
        }
    return BDS_OK;
}

static inline double cabs2(double2 v) { return std::hypot(v.x, v.y); }

// results(bin, lag) in f64 from the per-component coherent sums (B2a/acquisition.m:208-209,
// B1C/acquisition.m:212,218-219)
static inline double combine(const AcqState &a, const double2 *v) {
    if (a.signal == BDS_SIGNAL_B2A) return cabs2(v[0]) + cabs2(v[1]);
    if (a.ncomp == 1) return cabs2(v[0]);
    return (cabs2(v[0]) * std::sqrt(11.0) + cabs2(v[1]) * std::sqrt(29.0)) / std::sqrt(40.0);
}

}  // namespace bds

namespace bds {
namespace {

constexpr int kExtraCap = 1 << 22;  // entries of the sieve's candidate list (wave-private pass) / overflow list (tile pass)
constexpr int kSamples = 32;        // launch pairs of a run bracketed by timing events

// events of one run; released on every exit path
struct EventPool {
    std::vector<hipEvent_t> all;
    hipError_t make(hipEvent_t *e, unsigned flags = 0) {
        const hipError_t rc_ = flags ? hipEventCreateWithFlags(e, flags) : hipEventCreate(e);
        if (rc_ == hipSuccess) all.push_back(*e);
        return rc_;
    }
    ~EventPool() {
        for (hipEvent_t e : all) (void)hipEventDestroy(e);
    }
};

// packed cell maximum -> (value, 0-based lag); nothing searched / nothing written: (-1, -1)
void unpack_cell(unsigned long long pk, float *v, int *lag) {
    if (pk == 0) {
        *v = -1.f, *lag = -1;
        return;
    }
    const uint32_t hi = (uint32_t)(pk >> 32);
    memcpy(v, &hi, sizeof(float));
    *lag = (int)~(uint32_t)(pk & 0xffffffffu);
}

// What a stage asks of bds_acq_run when the sieve cannot be trusted with the storage / kernels it ran on (never returned
// through the C ABI): redo the call with fp32 storage, or on the run-time-plan kernels (one record per tile, first-index ties)
//  * a non-finite row maximum (an fp16 value overflowed; Parseval bounds every fp16 value by sqrt(L) x its unit RMS < 2^11, so
//    int8 input cannot get here -- kept for non-finite f64 input, exercised by a test hook);
//  * the f64 peak disagrees with the sieve's maximum beyond kDelta / 2: the error model does not hold for this input;
//  * the candidate list ran over: at the fp16 tolerance (a nearly flat surface: an interferer 40 dB above the noise) fp32
//    storage first; with fp32 storage (massive exact ties, e.g. an all-zero block) the run-time-plan kernels.
constexpr int kRedoFp32 = -1000, kRedoPlain = -1001;

// One bds_acq_run attempt: inputs, the quantities its stages share, and the stages in call order.
struct AcqRun {
    bds_ctx *ctx;
    AcqState &a;
    const bds_settings *s;  // effective settings (after the resampling branch)
    std::vector<int> prns;
    double *carrFreq, *codePhase, *peakMetric;
    int32_t *detected;
    std::string why;  // reason of a kRedo* return

    int P = 0, D = 0, G = 0, ncomp = 0;
    double f0 = 0, kDelta = 0;
    float w0 = 1.f, w1 = 1.f;
    bool fsearch = false, wcols = false, multiprn = false, overlap = false;
    bool dev_refined = false;  // the refinement ran as the device chain
    size_t elem = 8;  // bytes of one stored complex value
    int PB = 1;
    long n_pairs_total = 0, cells_per_pair = 0;
    SieveOut so{};
    // timing
    EventPool evp;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    hipEvent_t sa[kSamples], sb[kSamples], sm[kSamples];
    hipEvent_t ev_rows[2] = {nullptr, nullptr}, ev_cols[2] = {nullptr, nullptr};
    int nsamp = 0;
    bool mids = false;  // the sampled pairs carry a mid event (fp32-arithmetic kernels)
    size_t half_bytes = 0;  // one group of cells in the inter-pass buffer
    // the sieve's output and the decisions made on it
    int n_extra = 0;
    std::vector<Extra> h_extra;
    std::vector<float> thr_of, max_of;
    std::vector<std::vector<Cell>> cells;
    std::vector<PrnResult> res;

    AcqRun(bds_ctx *c, AcqState &st_, const bds_settings *s_) : ctx(c), a(st_), s(s_) {}
    hipStream_t stream() const { return st(ctx); }
    double bin_freq(int b) const { return f0 + s->acqStep * (double)b; }
    int redo(int code, const char *reason) {
        why = reason;
        return code;
    }

    int setup();        // sizes, work buffers, events, storage scales, the sieve's lists
    int forward_all();  // carrier wipe-off + forward transform of every Doppler bin
    void launch_cells(int prn, int b0, int nb, Rec *recs, int lo1, int hi1, int lo2, int hi2, int cell0, hipEvent_t mid, int buf = -1);
    void launch_list(int ncells, Rec *recs, const CellList &cl, int cell0, hipEvent_t mid);
    int search();          // all (PRN, bin) cells: row pass + column pass per group
    int collect();         // row maxima + list to the host; can the sieve be trusted?
    int refine();          // candidates -> f64 coherent sums -> peak, bin, code phase per PRN
    int metric_b1c_sigpower();
    int metric_b1c();      // GLRT normaliser
    int second_peak_b2a(); // second peak of the winning bin
    int fine_search();     // threshold + fine-Doppler search, results
    int finish();          // timing record
    // the same decisions as collect() .. fine_search() as one chain of launches with a single download (bds_acq_refine.h)
    bool device_refine_ok() const;
    int refine_device();
};

int AcqRun::setup() {
    Plan2D &pl = a.plan;
    const Tuning &tune = ctx->tune;
    int rc;
    D = (int)m_round(s->acqSearchBand * 2 / s->acqStep) + 1;  // numberOfFrqBins :150
    f0 = s->IF - s->acqSearchBand;                            // frqBins :190-191
    a.D = D;
    P = (int)prns.size();
    ncomp = a.ncomp;
    pick_group(a, *s);
    G = a.group;
    if ((rc = ensure(ctx, &a.d_Bw, &a.bw_cap, bw_batches(a) * (size_t)pl.L))) return rc;
    if ((rc = ensure(ctx, &a.d_Xs, &a.xs_cap, (size_t)D * pl.L))) return rc;

    BDS_HIP(ctx, evp.make(&ev0));
    BDS_HIP(ctx, evp.make(&ev1));
    BDS_HIP(ctx, evp.make(&ev2));
    BDS_HIP(ctx, evp.make(&ev3));
    for (int i = 0; i < kSamples; ++i) {
        BDS_HIP(ctx, evp.make(&sa[i]));
        BDS_HIP(ctx, evp.make(&sb[i]));
        BDS_HIP(ctx, evp.make(&sm[i]));
    }
    BDS_HIP(ctx, hipEventRecord(ev0, stream()));

    // ---- storage scales (fp16 mode): powers of two from exact sums of the block ----------
    a.sX = a.sB = 1.f;
    if (a.sums_N != a.N || a.sums_next != a.n_ext) ext_sums(a);  // settings of this run changed N after bds_acq_load
    if (a.half) {
        // |X[k]| <= sum|x|  -> keep the stored spectrum below 2^15
        a.sX = (float)std::exp2(std::floor(std::log2(32768.0 / std::max(1.0, a.sum_abs_ext))));
        // inter-pass values: rms = X_rms * C_rms / L * sqrt(L2) (Parseval); allow 64 x rms
        const double b_rms = std::sqrt(a.sum_sq_ext) * std::sqrt((double)a.X) / (double)pl.L * std::sqrt((double)pl.L2);
        a.sB = (float)std::exp2(std::floor(std::log2(32768.0 / (64.0 * std::max(1e-30, b_rms) * a.sX * a.sC))));
    }

    // ---- the search's knobs and lists ------------------------------------------------------
    w0 = w1 = 1.f;
    if (a.signal == BDS_SIGNAL_B1C && ncomp == 2) {
        w0 = (float)(std::sqrt(11.0) / std::sqrt(40.0));
        w1 = (float)(std::sqrt(29.0) / std::sqrt(40.0));
    }
    {   // undo the storage scales in the final magnitude weights
        const float inv = 1.0f / (a.sX * a.sC * a.sB);
        w0 *= inv;
        w1 *= inv;
    }
    fsearch = (pl.fast || (pl.small && a.half)) && !a.no_fast_search;  // fp32-arithmetic specialised kernels (default)
    // sieve tolerance: every lag within kDelta of a PRN's maximum is re-evaluated in f64.  fp32 storage (with the fp32 carrier / twiddle
    // rotations of the forward pass, round 4) errs by 5.3e-7 of the PRN maximum at worst against the f64 oracle (tools/sieve_error.py,
    // profiles/r05_sieve_error_small.txt: 19x inside its kDelta / 2 = 1e-5; checked at run time like the fp16 mode).  fp16 storage: three roundings lie between the exact value and the sieve's (signal spectrum, code spectrum,
    // inter-pass buffer; 2^-11 relative each).  On noise-like spectra they average out -- 2.5e-4 of the PRN maximum at worst
    // over 63 x 201 rows -- but a spectrum dominated by ONE line (a CW interferer) carries them coherently: 6.4e-4 / 7.9e-4
    // measured at J/N = +20 / +40 dB (tools/sieve_stress.py, profiles/r04_sieve_error.txt), bounded by 3 x 2^-11 = 1.46e-3.
    // kDelta / 2 = 2e-3 lies above that bound and 2.5x above the worst measured case (round 3 used 2e-3: 1.3x).
    kDelta = tune.kdelta > 0 ? tune.kdelta : a.half ? 4e-3 : 2e-5;
    {
        size_t cap = a.extra_cap;
        if ((rc = ensure(ctx, &a.d_extra, &cap, (size_t)kExtraCap))) return rc;
        a.extra_cap = cap;
        if (!a.d_extra_count) BDS_HIP(ctx, hipMalloc((void **)&a.d_extra_count, sizeof(int)));
        BDS_HIP(ctx, hipMemsetAsync(a.d_extra_count, 0, sizeof(int), stream()));
    }
    // wave-private column pass (default for the fp32-arithmetic search): per-cell packed maxima and per-PRN running
    // bounds instead of per-tile records
    // (the 256-point plans keep the tile kernel unless forced with BDS_ACQ_WCOLS=1: a workgroup's share of such a tile is
    //  8 points per lane and the per-workgroup constants and barriers dominate -- measured at cfg2 2.18 vs 1.39 ms per launch)
    wcols = fsearch && (pl.small || (tune.wcols != 0 && (pl.L1 != 256 || tune.wcols > 0)));
    so = SieveOut{nullptr, a.d_extra, a.d_extra_count, kExtraCap, 0, (float)(1.0 - kDelta)};
    if (wcols) {
        if ((rc = ensure(ctx, &a.d_cellmax, &a.cellmax_cap, (size_t)std::max(P, 1) * D))) return rc;
        if ((rc = ensure(ctx, &a.d_lb, &a.lb_cap, (size_t)std::max(P, 1)))) return rc;
        BDS_HIP(ctx, hipMemsetAsync(a.d_cellmax, 0, sizeof(unsigned long long) * (size_t)std::max(P, 1) * D, stream()));
        BDS_HIP(ctx, hipMemsetAsync(a.d_lb, 0, sizeof(float) * (size_t)std::max(P, 1), stream()));
        so.cellmax = a.d_cellmax;
        so.lb = a.d_lb;
        so.lb_div = D;
    } else {  // per-tile records + per-row reduction (tile kernel, run-time-plan kernels)
        if ((rc = ensure(ctx, &a.d_recs, &a.recs_cap, (size_t)std::max(P, 1) * D * pl.ntiles))) return rc;
        if (a.rows_cap < (size_t)P * D) {
            if (a.d_rowmax) (void)hipFree(a.d_rowmax), a.d_rowmax = nullptr;
            if (a.d_rowarg) (void)hipFree(a.d_rowarg), a.d_rowarg = nullptr;
            size_t d0 = 0, d1 = 0;
            if ((rc = ensure(ctx, &a.d_rowmax, &d0, (size_t)P * D))) return rc;
            if ((rc = ensure(ctx, &a.d_rowarg, &d1, (size_t)P * D))) return rc;
            a.rows_cap = (size_t)P * D;
        }
        so.recs = a.d_recs;
    }
    elem = a.half ? 4 : 8;
    // Small Doppler grids (B2a: 26 bins): one launch pair carries the whole rows of several PRNs through
    // a cell list, so that the grids fill the chip and a row workgroup still walks one PRN's bins.
    multiprn = fsearch && (D <= 104 || tune.multi_any) && P > 1 && !tune.nomulti;
    // (measured at cfg2, 63 PRNs x 26 bins: 104 cells per pair 3.9 ms, 208 -> 3.6, 416 -> 3.2, 832 -> 3.1, all 1638 -> 3.0;
    //  the fused fp16 chain of round 2 4.0); the work buffer is capped at 8 GiB
    const long pb_cap = std::max<long>(1, (long)(tune.pbcap_gb * 1073741824.0 / ((double)ncomp * (double)pl.L * (double)elem)));
    const long pb_cells = tune.pbcells ? tune.pbcells : pb_cap;
    PB = multiprn ? (int)std::min<long>(P, std::max<long>(2, std::min(pb_cells, pb_cap) / D)) : 1;
    n_pairs_total = (long)P * ((D + G - 1) / G);
    cells_per_pair = G;
    if (multiprn) n_pairs_total = (P + PB - 1) / PB, cells_per_pair = (long)PB * D;
    // Overlapped passes (BDS_ACQ_OVERLAP=1, fp32-arithmetic kernels): group k's column pass runs on a second stream beside
    // group k+1's row pass, the two working in different halves of the inter-pass buffer.
    overlap = fsearch && tune.overlap && !multiprn;
    if (overlap)
        for (int i = 0; i < 2; ++i) {
            BDS_HIP(ctx, evp.make(&ev_rows[i], hipEventDisableTiming));
            BDS_HIP(ctx, evp.make(&ev_cols[i], hipEventDisableTiming));
        }
    half_bytes = (size_t)G * ncomp * pl.L * elem;
    return BDS_OK;
}

int AcqRun::forward_all() {
    Plan2D &pl = a.plan;
    const int chunk = (int)bw_batches(a);
    for (int b0 = 0; b0 < D; b0 += chunk) {
        const int nb = std::min(chunk, D - b0);
        SignalLoader ld{a.sview(), a.N, a.n_ext, f0, s->acqStep, 1.0 / a.fs, b0};
        float2 *xs_dst = a.half ? (float2 *)((__half2 *)a.d_Xs + (size_t)b0 * pl.L) : a.d_Xs + (size_t)b0 * pl.L;
        if (int rc = forward(ctx, a, ld, nb, xs_dst, pl.L, 0, a.sX)) return rc;
    }
    BDS_HIP(ctx, hipEventRecord(ev1, stream()));
    return BDS_OK;
}

// one group of cells of one PRN (consecutive bins b0 .. b0+nb-1, or one bin with lag ranges): both passes on the
// main stream; cell0 = run-wide index of the first cell (list bookkeeping)
void AcqRun::launch_cells(int prn, int b0, int nb, Rec *recs, int lo1, int hi1, int lo2, int hi2, int cell0, hipEvent_t mid, int buf) {
    Plan2D &pl = a.plan;
    const hipStream_t s_main = stream();
    const size_t cs_off = (size_t)a.cs_slot[prn] * ncomp * pl.L;
    SieveOut so1 = so;
    so1.recs = recs;
    so1.cell0 = cell0;
    so1.mid = mid;
    if (mid && fsearch) mids = true;
    void *const Bw_ = buf > 0 ? (void *)((char *)a.d_Bw + half_bytes) : (void *)a.d_Bw;
    if (buf >= 0) {
        so1.cols_stream = (hipStream_t)ctx->stream2;
        so1.ev_rows = ev_rows[buf];
        so1.ev_cols = ev_cols[buf];
    }
    if (fsearch && a.half) {
        const void *Ch = (const __half2 *)a.d_Cs + cs_off;
        if (ncomp == 2)
            launch_fast_f<2, __half2>(ctx, s_main, pl, a.d_Xs, nb, b0, Ch, Bw_, a.sB, w0, w1, lo1, hi1, lo2, hi2, so1);
        else
            launch_fast_f<1, __half2>(ctx, s_main, pl, a.d_Xs, nb, b0, Ch, Bw_, a.sB, w0, w1, lo1, hi1, lo2, hi2, so1);
    } else if (fsearch) {
        const void *Cf = a.d_Cs + cs_off;
        if (ncomp == 2)
            launch_fast_f<2, float2>(ctx, s_main, pl, a.d_Xs, nb, b0, Cf, Bw_, a.sB, w0, w1, lo1, hi1, lo2, hi2, so1);
        else
            launch_fast_f<1, float2>(ctx, s_main, pl, a.d_Xs, nb, b0, Cf, Bw_, a.sB, w0, w1, lo1, hi1, lo2, hi2, so1);
    } else {
        const float2 *Cs = a.d_Cs + cs_off;
        dim3 gr(pl.L1, nb), gc(pl.ntiles, nb);
        if (ncomp == 2) {
            hipLaunchKernelGGL(k_rows_inv<2>, gr, dim3(pl.nt_rows), pl.lds_rows, s_main, pl.p2, pl.twl,
                               (const float2 *)a.d_Xs, pl.L, b0, Cs, a.d_Bw);
            hipLaunchKernelGGL(k_cols_inv_max<2>, gc, dim3(pl.nt_cols), pl.lds_cols, s_main, pl.p1, pl.L2, pl.logT,
                               pl.Spad, (const float2 *)a.d_Bw, pl.L, w0, w1, lo1, hi1, lo2, hi2, recs, pl.ntiles);
        } else {
            hipLaunchKernelGGL(k_rows_inv<1>, gr, dim3(pl.nt_rows), pl.lds_rows, s_main, pl.p2, pl.twl,
                               (const float2 *)a.d_Xs, pl.L, b0, Cs, a.d_Bw);
            hipLaunchKernelGGL(k_cols_inv_max<1>, gc, dim3(pl.nt_cols), pl.lds_cols, s_main, pl.p1, pl.L2, pl.logT,
                               pl.Spad, (const float2 *)a.d_Bw, pl.L, w0, w1, lo1, hi1, lo2, hi2, recs, pl.ntiles);
        }
    }
}

// cells described by a list (whole rows of several PRNs, or one (PRN, winning bin) cell per PRN)
void AcqRun::launch_list(int ncells, Rec *recs, const CellList &cl, int cell0, hipEvent_t mid) {
    Plan2D &pl = a.plan;
    const hipStream_t s_main = stream();
    SieveOut so1 = so;
    so1.recs = recs;
    so1.cell0 = cell0;
    so1.mid = mid;
    if (mid) mids = true;
    const int hi1 = cl.rng ? -1 : (int)a.N - 1, lo2 = cl.rng ? 0 : 1, hi2 = cl.rng ? -1 : 0;
    if (a.half) {
        if (ncomp == 2)
            launch_fast_f<2, __half2>(ctx, s_main, pl, a.d_Xs, ncells, 0, a.d_Cs, a.d_Bw, a.sB, w0, w1, 0, hi1, lo2, hi2, so1, cl);
        else
            launch_fast_f<1, __half2>(ctx, s_main, pl, a.d_Xs, ncells, 0, a.d_Cs, a.d_Bw, a.sB, w0, w1, 0, hi1, lo2, hi2, so1, cl);
    } else {
        if (ncomp == 2)
            launch_fast_f<2, float2>(ctx, s_main, pl, a.d_Xs, ncells, 0, a.d_Cs, a.d_Bw, a.sB, w0, w1, 0, hi1, lo2, hi2, so1, cl);
        else
            launch_fast_f<1, float2>(ctx, s_main, pl, a.d_Xs, ncells, 0, a.d_Cs, a.d_Bw, a.sB, w0, w1, 0, hi1, lo2, hi2, so1, cl);
    }
}

int AcqRun::search() {
    Plan2D &pl = a.plan;
    const hipStream_t s_main = stream();
    int rc;
    const long sample_every = std::max<long>(1, n_pairs_total / kSamples);
    long pair_idx = 0, group_idx = 0;
    if (multiprn) {
        // float2-sized elements the PB*D cells of one launch pair occupy
        const size_t need = ((size_t)PB * D * ncomp * elem + 7) / 8;
        if ((rc = ensure(ctx, &a.d_Bw, &a.bw_cap, std::max(need, bw_batches(a)) * (size_t)pl.L))) return rc;
        const size_t nc_ = (size_t)P * D;
        std::vector<int> h_bin(nc_);
        std::vector<long> h_cs(nc_);
        for (int pi = 0; pi < P; ++pi)
            for (int b = 0; b < D; ++b) {
                h_bin[(size_t)pi * D + b] = b;
                h_cs[(size_t)pi * D + b] = (long)a.cs_slot[prns[pi]] * ncomp * pl.L;
            }
        if ((rc = ensure(ctx, &a.d_cells, &a.cells_cap, (sizeof(long) + sizeof(int)) * nc_ + 64))) return rc;
        long *d_cs = (long *)a.d_cells;
        int *d_bin = (int *)(d_cs + nc_);
        BDS_HIP(ctx, hipMemcpyAsync(d_cs, h_cs.data(), sizeof(long) * nc_, hipMemcpyHostToDevice, s_main));
        BDS_HIP(ctx, hipMemcpyAsync(d_bin, h_bin.data(), sizeof(int) * nc_, hipMemcpyHostToDevice, s_main));
        for (int pi0 = 0; pi0 < P; pi0 += PB, ++pair_idx) {
            const int np_ = std::min(PB, P - pi0);
            CellList cl;
            cl.bin = d_bin + (size_t)pi0 * D;
            cl.cs = d_cs + (size_t)pi0 * D;
            cl.gc = D;
            const bool sample = np_ == PB && (pair_idx % sample_every) == 0 && nsamp < kSamples;
            if (sample) BDS_HIP(ctx, hipEventRecord(sa[nsamp], s_main));
            launch_list(np_ * D, wcols ? nullptr : a.d_recs + (size_t)pi0 * D * pl.ntiles, cl, pi0 * D, sample ? sm[nsamp] : nullptr);
            if (sample) BDS_HIP(ctx, hipEventRecord(sb[nsamp++], s_main));
        }
    } else {
        for (int pi = 0; pi < P; ++pi) {
            for (int b0 = 0; b0 < D; b0 += G, ++pair_idx, ++group_idx) {
                const int nb = std::min(G, D - b0);
                const int buf = overlap ? (int)(group_idx & 1) : -1;
                // the row pass of group k re-uses the buffer half the column pass of group k-2 read
                if (overlap && group_idx >= 2) BDS_HIP(ctx, hipStreamWaitEvent(s_main, ev_cols[buf], 0));
                const bool sample = !overlap && nb == G && (pair_idx % sample_every) == 0 && nsamp < kSamples;
                if (sample) BDS_HIP(ctx, hipEventRecord(sa[nsamp], s_main));
                launch_cells(prns[pi], b0, nb, wcols ? nullptr : a.d_recs + ((size_t)pi * D + b0) * pl.ntiles, 0, (int)a.N - 1, 1, 0,
                             pi * D + b0, sample ? sm[nsamp] : nullptr, buf);
                if (sample) BDS_HIP(ctx, hipEventRecord(sb[nsamp++], s_main));
            }
        }
        if (overlap)  // join: everything after this is ordered on the main stream again
            for (int i = 0; i < 2 && i < group_idx; ++i) BDS_HIP(ctx, hipStreamWaitEvent(s_main, ev_cols[i], 0));
    }
    BDS_HIP(ctx, hipGetLastError());
    if (!wcols)
        hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)(P * D)), dim3(256), 0, s_main, (const Rec *)a.d_recs,
                           pl.ntiles, pl.ntiles, a.d_rowmax, a.d_rowarg);
    BDS_HIP(ctx, hipEventRecord(ev2, s_main));
    return BDS_OK;
}

int AcqRun::collect() {
    const Tuning &tune = ctx->tune;
    a.h_rowmax.resize((size_t)P * D);
    a.h_rowarg.resize((size_t)P * D);
    n_extra = 0;
    std::vector<unsigned long long> h_cellmax(wcols ? (size_t)P * D : 0);
    if (wcols) {
        BDS_HIP(ctx, hipMemcpyAsync(h_cellmax.data(), a.d_cellmax, sizeof(unsigned long long) * P * D, hipMemcpyDeviceToHost, stream()));
    } else {
        BDS_HIP(ctx, hipMemcpyAsync(a.h_rowmax.data(), a.d_rowmax, sizeof(float) * P * D, hipMemcpyDeviceToHost, stream()));
        BDS_HIP(ctx, hipMemcpyAsync(a.h_rowarg.data(), a.d_rowarg, sizeof(int) * P * D, hipMemcpyDeviceToHost, stream()));
    }
    BDS_HIP(ctx, hipMemcpyAsync(&n_extra, a.d_extra_count, sizeof(int), hipMemcpyDeviceToHost, stream()));
    BDS_HIP(ctx, hipStreamSynchronize(stream()));
    for (size_t i = 0; i < h_cellmax.size(); ++i) unpack_cell(h_cellmax[i], &a.h_rowmax[i], &a.h_rowarg[i]);
    a.run_prns = prns;
    a.last.clear();
    {
        bool bad = false;
        for (float v : a.h_rowmax) bad = bad || !std::isfinite(v);
        if (bad && tune.verbose) {
            int nbad = 0;
            for (float v : a.h_rowmax) nbad += !std::isfinite(v);
            fprintf(stderr, "[bds] %d of %zu row maxima are not finite; first rows:", nbad, a.h_rowmax.size());
            for (size_t i = 0; i < std::min<size_t>(8, a.h_rowmax.size()); ++i) fprintf(stderr, " %g", a.h_rowmax[i]);
            fprintf(stderr, "  (sX %g sC %g sB %g)\n", a.sX, a.sC, a.sB);
        }
        if (a.half && ((bad && !tune.no_selfcheck) || tune.test_force_fallback)) return redo(kRedoFp32, bad ? "non-finite row maximum" : "test hook");
        if (n_extra > kExtraCap && a.half) return redo(kRedoFp32, "overflow list of the sieve ran over at the fp16-storage tolerance");
        if (n_extra > kExtraCap && !a.no_fast_search) return redo(kRedoPlain, "overflow list of the sieve ran over");
    }
    h_extra.resize((size_t)std::min(n_extra, kExtraCap));
    if (!h_extra.empty())
        BDS_HIP(ctx, hipMemcpyAsync(h_extra.data(), a.d_extra, sizeof(Extra) * h_extra.size(), hipMemcpyDeviceToHost, stream()));
    a.n_extra_last = n_extra;
    return BDS_OK;
}

// ---- f64 refinement of the sieve's candidates ---------------------------------------
int AcqRun::refine() {
    Plan2D &pl = a.plan;
    const Tuning &tune = ctx->tune;
    int rc;
    cells.assign(P, {});
    std::vector<CorrJob> jobs;
    // only the per-workgroup records of rows that reach the tolerance band travel to the host
    // (the full record array is P*D*tiles*8 B: 104 MB at the B1C config); the wave-private pass keeps no tile records: its list is complete
    thr_of.assign(P, 0.f);
    max_of.assign(P, 0.f);
    std::map<std::pair<int, int>, size_t> row_at;
    std::vector<Rec> h_recs;
    {
        std::vector<std::pair<int, int>> rows;
        for (int pi = 0; pi < P; ++pi) {
            float M = -1.f;
            for (int b = 0; b < D; ++b) M = std::max(M, a.h_rowmax[(size_t)pi * D + b]);
            max_of[pi] = M;
            thr_of[pi] = (float)((1.0 - kDelta) * (double)M);
            for (int b = 0; b < D && !wcols; ++b)
                if (!(a.h_rowmax[(size_t)pi * D + b] < thr_of[pi])) rows.push_back({pi, b});
        }
        const size_t all = (size_t)P * D * pl.ntiles;
        if (wcols) {
            // nothing to fetch
        } else if (all * sizeof(Rec) <= (16u << 20)) {  // small grid (B2a): one copy beats many row copies
            h_recs.resize(all);
            BDS_HIP(ctx, hipMemcpyAsync(h_recs.data(), a.d_recs, sizeof(Rec) * all, hipMemcpyDeviceToHost, stream()));
            for (auto &r : rows) row_at[r] = ((size_t)r.first * D + r.second) * pl.ntiles;
        } else {
            h_recs.resize(rows.size() * (size_t)pl.ntiles);
            for (size_t r = 0; r < rows.size(); ++r) {
                row_at[rows[r]] = r * (size_t)pl.ntiles;
                BDS_HIP(ctx, hipMemcpyAsync(&h_recs[r * (size_t)pl.ntiles],
                                            a.d_recs + ((size_t)rows[r].first * D + rows[r].second) * pl.ntiles,
                                            sizeof(Rec) * pl.ntiles, hipMemcpyDeviceToHost, stream()));
            }
        }
        BDS_HIP(ctx, hipStreamSynchronize(stream()));  // (also: h_extra has arrived)
    }
    {
        std::vector<std::set<Cell>> cs(P);
        // (rounds 1-3 also refined the +-1 bin / +-1 lag neighbours of every candidate -- nine f64 sums per candidate.  The
        //  completeness argument does not use them: the true maximum's sieve value is within kDelta / 2 of it, hence within
        //  kDelta of the sieve maximum, hence on the list itself.  BDS_ACQ_NEIGH=1 brings them back.)
        const int nb_r = tune.neigh;
        auto add = [&](int pi, int b, long lag) {
            for (int db = -nb_r; db <= nb_r; ++db)
                for (int dl = -nb_r; dl <= nb_r; ++dl) {
                    const int bb = b + db;
                    const long ll = lag + dl;
                    if (bb >= 0 && bb < D && ll >= 0 && ll < a.N) cs[pi].insert(Cell{bb, ll});
                }
        };
        for (int pi = 0; pi < P; ++pi) {
            const float thr = thr_of[pi];
            for (int b = 0; b < D && !wcols; ++b) {
                if (a.h_rowmax[(size_t)pi * D + b] < thr) continue;
                const Rec *rr = &h_recs[row_at[{pi, b}]];
                for (int t = 0; t < pl.ntiles; ++t)
                    if (rr[t].lag >= 0 && !(rr[t].v < thr)) add(pi, b, rr[t].lag);
            }
        }
        // lags the column pass put on its list (wave-private pass: every candidate; tile pass: those beside their tile's record)
        for (const Extra &e : h_extra) {
            const int pi = e.cell / D, b = e.cell % D;
            if (pi >= 0 && pi < P && e.lag >= 0 && !(e.v < thr_of[pi])) add(pi, b, e.lag);
        }
        a.last_cands.clear();
        for (int pi = 0; pi < P; ++pi) {
            cells[pi].assign(cs[pi].begin(), cs[pi].end());
            auto &lc = a.last_cands[prns[pi]];
            for (const Cell &c : cells[pi]) lc.push_back({c.b, c.lag});
            for (const Cell &c : cells[pi])
                for (int comp = 0; comp < ncomp; ++comp) {
                    CorrJob j{};
                    j.start = c.lag;
                    j.len = a.X;
                    j.freq = bin_freq(c.b);
                    j.mean = 0;
                    j.slot = (prns[pi] - 1) * 2 + comp;
                    j.circ = 1;
                    j.mode = 0;
                    jobs.push_back(j);
                }
        }
    }
    std::vector<double2> jout;
    if ((rc = run_jobs(ctx, a, *s, jobs, jout))) return rc;
    res.assign(P, PrnResult{});
    size_t k = 0;
    for (int pi = 0; pi < P; ++pi) {
        double best = -1;
        Cell bc{0, 0};
        for (const Cell &c : cells[pi]) {
            const double v = combine(a, &jout[k]);
            k += ncomp;
            // ties: first row / first column, as MATLAB max does (acquisition.m:218-221)
            if (v > best || (v == best && (c.b < bc.b || (c.b == bc.b && c.lag < bc.lag)))) best = v, bc = c;
        }
        res[pi].peak = best;
        res[pi].fbin = bc.b + 1;
        res[pi].codePhase = bc.lag + 1;
        // The sieve's maximum must agree with the f64 value to well inside the tolerance band it was
        // searched with; otherwise its error model does not hold for this input: redo with fp32 storage.
        // (round 5: fp32 storage on the specialised kernels is checked the same way against ITS tolerance -- its forward pass
        //  rotates the carrier in fp32 since round 4 -- and falls back to the run-time-plan kernels)
        if ((a.half || (fsearch && !a.no_fast_search)) && !tune.no_selfcheck && !cells[pi].empty() &&
            std::fabs(best - (double)max_of[pi]) > 0.5 * kDelta * best) {
            char msg[160];
            snprintf(msg, sizeof(msg), "PRN %d: sieve maximum %.9g vs f64 %.9g (rel %.3g > %.3g)", prns[pi], (double)max_of[pi], best,
                     std::fabs(best - (double)max_of[pi]) / best, 0.5 * kDelta);
            return redo(a.half ? kRedoFp32 : kRedoPlain, msg);
        }
    }
    return BDS_OK;
}

// sigPower = sqrt(var(sig(1:X)) * X), unbiased variance (B1C/acquisition.m:150)
// (complex input: var = sum |x - mean|^2 / (X-1), as MATLAB's var of a complex vector)
int AcqRun::metric_b1c_sigpower() {
    // (a property of the loaded block and X: a million-term host sum, kept across calls -- it was ~1.5 ms of every run)
    if (a.sigpower_X != a.X) {
        const double mean = (a.h_prefix[a.X] - a.h_prefix[0]) / (double)a.X;
        const double mean_q = a.cplx ? (a.h_prefix_q[a.X] - a.h_prefix_q[0]) / (double)a.X : 0.0;
        long double acc = 0;
        for (long i = 0; i < a.X; ++i) {
            const double d = a.h_re[(size_t)i] - mean;
            const double dq = a.cplx ? a.h_im[(size_t)i] - mean_q : 0.0;
            acc += (long double)(d * d + dq * dq);
        }
        const double var = (double)(acc / (long double)(a.X - 1));
        a.sigpower = std::sqrt(var * (double)a.X);
        a.sigpower_X = a.X;
    }
    return BDS_OK;
}

int AcqRun::metric_b1c() {
    if (int rc = metric_b1c_sigpower()) return rc;
    const double sigPower = a.sigpower;
    for (int pi = 0; pi < P; ++pi) {
        res[pi].denom = sigPower;
        if (res[pi].codePhase + a.spc - 1 > a.n_samples) res[pi].codePhase -= a.spc;  // :239-241
    }
    return BDS_OK;
}

// second peak in the winning bin, outside +-2 chips and within +-1 code (B2a/acquisition.m:224-249)
// (one cell per PRN, a single round of workgroups: the tile kernel with its per-tile records serves this pass; the
//  wave-private kernel's running bounds have nothing to run on)
int AcqRun::second_peak_b2a() {
    Plan2D &pl = a.plan;
    const int nb_r = ctx->tune.neigh;
    int rc;
    const bool small = pl.small && fsearch;  // the 80 x 4096 plan has no tile kernel: its column pass reports as in the search
    so.cellmax = nullptr;
    so.lb = nullptr;
    const long s2c = (long)std::ceil(s->samplingFreq / s->codeFreqBasis) * 2;  // samples2CodeChip :137
    std::vector<std::array<long, 4>> rng(P);
    if (small) {  // cell = PRN index: one packed maximum and one running bound per PRN
        BDS_HIP(ctx, hipMemsetAsync(a.d_cellmax, 0, sizeof(unsigned long long) * (size_t)std::max(P, 1), stream()));
        BDS_HIP(ctx, hipMemsetAsync(a.d_lb, 0, sizeof(float) * (size_t)std::max(P, 1), stream()));
        so.cellmax = a.d_cellmax;
        so.lb = a.d_lb;
        so.lb_div = 1;
        so.recs = nullptr;
    } else {
        if ((rc = ensure(ctx, &a.d_recs, &a.recs_cap, (size_t)std::max(P, 1) * pl.ntiles))) return rc;
        so.recs = a.d_recs;
    }
    // specialised kernels: all PRNs' (PRN, winning bin) cells in one launch pair through a cell list
    // (63 tiny launch pairs were ~1 ms of the 2.7 ms refinement at cfg2)
    const size_t cap_cells = a.bw_cap / (size_t)pl.L * 8 / elem / (size_t)ncomp;  // cells the work buffer holds
    const bool batched = fsearch && (size_t)P <= cap_cells;
    BDS_HIP(ctx, hipMemsetAsync(a.d_extra_count, 0, sizeof(int), stream()));  // overflow list of this pass: cell = PRN index
    std::vector<int> h_bin(P);
    std::vector<long> h_cs(P);
    std::vector<int4> h_rng(P);
    for (int pi = 0; pi < P; ++pi) {
        const long cp = res[pi].codePhase;
        const long e1 = cp - s2c, e2 = cp + s2c, e3 = cp - a.spc + s2c, e4 = cp + a.spc - s2c;
        long lo1 = 1, hi1 = 0, lo2 = 1, hi2 = 0;  // 1-based inclusive, empty when lo > hi
        if (e1 >= 1) lo1 = std::max<long>(1, e3), hi1 = e1;
        if (e2 < a.N) lo2 = e2, hi2 = std::min<long>(e4, a.N);
        rng[pi] = {lo1 - 1, hi1 - 1, lo2 - 1, hi2 - 1};  // 0-based
        if (hi1 < lo1 && hi2 < lo2)
            return fail(ctx, BDS_ERR_ARG, "PRN %d: empty second-peak range (acquisition.m:248 would fail)", prns[pi]);
        h_bin[pi] = res[pi].fbin - 1;
        h_cs[pi] = (long)a.cs_slot[prns[pi]] * ncomp * pl.L;
        h_rng[pi] = make_int4((int)rng[pi][0], (int)rng[pi][1], (int)rng[pi][2], (int)rng[pi][3]);
        if (!batched)
            launch_cells(prns[pi], res[pi].fbin - 1, 1, small ? nullptr : a.d_recs + (size_t)pi * pl.ntiles, (int)rng[pi][0],
                         (int)rng[pi][1], (int)rng[pi][2], (int)rng[pi][3], pi, nullptr);
    }
    if (batched && P > 0) {
        const size_t nb_ = sizeof(int) * P + sizeof(long) * P + sizeof(int4) * P + 64;
        if ((rc = ensure(ctx, &a.d_cells, &a.cells_cap, nb_))) return rc;
        int4 *d_rng = (int4 *)a.d_cells;                       // 16-byte aligned first
        long *d_cs = (long *)(d_rng + P);
        int *d_bin = (int *)(d_cs + P);
        BDS_HIP(ctx, hipMemcpyAsync(d_rng, h_rng.data(), sizeof(int4) * P, hipMemcpyHostToDevice, stream()));
        BDS_HIP(ctx, hipMemcpyAsync(d_cs, h_cs.data(), sizeof(long) * P, hipMemcpyHostToDevice, stream()));
        BDS_HIP(ctx, hipMemcpyAsync(d_bin, h_bin.data(), sizeof(int) * P, hipMemcpyHostToDevice, stream()));
        const CellList cl{d_bin, d_cs, d_rng};
        launch_list(P, small ? nullptr : a.d_recs, cl, 0, nullptr);
    }
    BDS_HIP(ctx, hipGetLastError());
    std::vector<Rec> r2(small ? 0 : (size_t)P * pl.ntiles);
    std::vector<unsigned long long> h_cm(small ? (size_t)P : 0);
    int n_extra2 = 0;
    if (small)
        BDS_HIP(ctx, hipMemcpyAsync(h_cm.data(), a.d_cellmax, sizeof(unsigned long long) * (size_t)P, hipMemcpyDeviceToHost, stream()));
    else
        BDS_HIP(ctx, hipMemcpyAsync(r2.data(), a.d_recs, sizeof(Rec) * r2.size(), hipMemcpyDeviceToHost, stream()));
    BDS_HIP(ctx, hipMemcpyAsync(&n_extra2, a.d_extra_count, sizeof(int), hipMemcpyDeviceToHost, stream()));
    BDS_HIP(ctx, hipStreamSynchronize(stream()));
    if (n_extra2 > kExtraCap && a.half) return redo(kRedoFp32, "overflow list of the second-peak pass ran over at the fp16-storage tolerance");
    if (n_extra2 > kExtraCap && !a.no_fast_search) return redo(kRedoPlain, "overflow list of the second-peak pass ran over");
    std::vector<Extra> h_extra2((size_t)std::min(n_extra2, kExtraCap));
    if (!h_extra2.empty()) {
        BDS_HIP(ctx, hipMemcpyAsync(h_extra2.data(), a.d_extra, sizeof(Extra) * h_extra2.size(), hipMemcpyDeviceToHost, stream()));
        BDS_HIP(ctx, hipStreamSynchronize(stream()));
    }

    std::vector<CorrJob> jobs;
    std::vector<std::vector<long>> lags(P);
    for (int pi = 0; pi < P; ++pi) {
        float M = -1.f;
        if (small) {
            int lag_unused;
            unpack_cell(h_cm[(size_t)pi], &M, &lag_unused);  // (the maximum itself is on the list, like every lag above the threshold)
        }
        for (int t = 0; t < pl.ntiles && !small; ++t) M = std::max(M, r2[(size_t)pi * pl.ntiles + t].v);
        const float thr = (float)((1.0 - kDelta) * (double)M);
        std::set<long> ls;
        auto inrange = [&](long l) {
            return (l >= rng[pi][0] && l <= rng[pi][1]) || (l >= rng[pi][2] && l <= rng[pi][3]);
        };
        for (int t = 0; t < pl.ntiles && !small; ++t) {
            const Rec &r = r2[(size_t)pi * pl.ntiles + t];
            if (r.lag < 0 || r.v < thr) continue;
            for (long dl = -nb_r; dl <= nb_r; ++dl)
                if (inrange(r.lag + dl)) ls.insert(r.lag + dl);
        }
        for (const Extra &e : h_extra2)
            if (e.cell == pi && e.lag >= 0 && !(e.v < thr))
                for (long dl = -nb_r; dl <= nb_r; ++dl)
                    if (inrange(e.lag + dl)) ls.insert(e.lag + dl);
        lags[pi].assign(ls.begin(), ls.end());
        for (long l : lags[pi])
            for (int comp = 0; comp < ncomp; ++comp) {
                CorrJob j{};
                j.start = l;
                j.len = a.X;
                j.freq = bin_freq(res[pi].fbin - 1);
                j.slot = (prns[pi] - 1) * 2 + comp;
                j.circ = 1;
                j.mode = 0;
                jobs.push_back(j);
            }
    }
    std::vector<double2> jout;
    if ((rc = run_jobs(ctx, a, *s, jobs, jout))) return rc;
    size_t k = 0;
    for (int pi = 0; pi < P; ++pi) {
        double second = -1;
        for (size_t i = 0; i < lags[pi].size(); ++i, k += ncomp) second = std::max(second, combine(a, &jout[k]));
        res[pi].denom = second;
    }
    return BDS_OK;
}

// ---- threshold + fine-Doppler search ---------------------------------------------------
int AcqRun::fine_search() {
    int rc;
    std::vector<CorrJob> jobs;
    std::vector<int> fine_of(P, -1);
    int nfine = 0;
    std::vector<std::vector<double>> fine_frq(P);
    for (int pi = 0; pi < P; ++pi) {
        PrnResult &r = res[pi];
        const double metric = r.peak / r.denom;  // :252 / B1C :235
        peakMetric[prns[pi] - 1] = metric;
        if (!(metric > s->acqThreshold)) continue;  // :255 / B1C :244
        r.detected = true;
        const double fb = bin_freq(r.fbin - 1);
        if (a.signal == BDS_SIGNAL_B1C) {
            nfine = (int)m_round(s->acqStep / 25) * 2 + 1;  // B1C/acquisition.m:267
            if (r.codePhase < 1 || r.codePhase - 1 + a.spc > a.n_samples)
                return fail(ctx, BDS_ERR_ARG, "PRN %d: fine-search block %ld..%ld outside longSignal (B1C/acquisition.m:253)",
                            prns[pi], r.codePhase, r.codePhase + a.spc - 1);
            const double mean = (a.h_prefix[r.codePhase - 1 + a.spc] - a.h_prefix[r.codePhase - 1]) / (double)a.spc;  // :254
            const double mean_q = a.cplx ? (a.h_prefix_q[r.codePhase - 1 + a.spc] - a.h_prefix_q[r.codePhase - 1]) / (double)a.spc : 0.0;
            for (int kf = 0; kf < nfine; ++kf) fine_frq[pi].push_back(fb - s->acqStep + 25.0 * kf);  // :282-283
            // jobs of one PRN: [component][chunk of up to kCorrFreqs frequencies]
            for (int comp = 0; comp < ncomp; ++comp)
                for (int k0 = 0; k0 < nfine; k0 += kCorrFreqs) {
                    CorrJob j{};
                    j.start = r.codePhase - 1;
                    j.len = a.spc;
                    j.mean = mean;
                    j.mean_q = mean_q;
                    j.slot = (prns[pi] - 1) * 2 + comp;
                    j.circ = 0;
                    j.mode = 0;
                    j.nf = std::min(kCorrFreqs, nfine - k0);
                    for (int f = 0; f < j.nf; ++f) j.fr[f] = fine_frq[pi][k0 + f];
                    j.freq = j.fr[0];
                    jobs.push_back(j);
                }
        } else {
            nfine = (int)m_round(s->acqStep / 25) + 1;  // B2a/acquisition.m:265
            const long nn = (long)s->fineNoncoh * a.spc;
            if (r.codePhase - 1 + nn > a.n_samples)
                return fail(ctx, BDS_ERR_ARG, "PRN %d: fine-search block %ld..%ld outside longSignal (B2a/acquisition.m:290)",
                            prns[pi], r.codePhase, r.codePhase + nn - 1);
            for (int kf = 0; kf < nfine; ++kf) fine_frq[pi].push_back(fb - s->acqStep / 2 + 25.0 * kf);  // :300-301
            // jobs of one PRN: [segment][component][chunk of up to kCorrFreqs frequencies]
            for (int seg = 0; seg < s->fineNoncoh; ++seg)
                for (int comp = 0; comp < 2; ++comp)
                    for (int k0 = 0; k0 < nfine; k0 += kCorrFreqs) {
                        CorrJob j{};
                        j.start = r.codePhase - 1 + (long)seg * a.spc;
                        j.len = a.spc;
                        j.code_k0 = (long)seg * a.spc;
                        j.slot = (prns[pi] - 1) * 2 + comp;
                        j.circ = 0;
                        j.mode = 1;
                        j.nf = std::min(kCorrFreqs, nfine - k0);
                        for (int f = 0; f < j.nf; ++f) j.fr[f] = fine_frq[pi][k0 + f];
                        j.freq = j.fr[0];
                        jobs.push_back(j);
                    }
        }
        fine_of[pi] = 1;
    }
    std::vector<double2> jout;
    if ((rc = run_jobs(ctx, a, *s, jobs, jout, true))) return rc;
    const int nchunk = (nfine + kCorrFreqs - 1) / kCorrFreqs;
    size_t job0 = 0;  // first job of the PRN
    for (int pi = 0; pi < P; ++pi) {
        if (fine_of[pi] < 0) continue;
        // sum of frequency kf of job group (seg, comp): jobs are laid out [seg][comp][chunk]
        auto at = [&](int seg, int comp, int ncomp_, int kf) {
            const size_t j = job0 + ((size_t)seg * ncomp_ + comp) * nchunk + kf / kCorrFreqs;
            return jout[j * kCorrFreqs + kf % kCorrFreqs];
        };
        double best = -1;
        int kbest = 0;
        for (int kf = 0; kf < nfine; ++kf) {
            double v;
            if (a.signal == BDS_SIGNAL_B1C) {
                v = cabs2(at(0, 0, ncomp, kf));
                if (ncomp == 2) v = (v * 11 + cabs2(at(0, 1, ncomp, kf)) * 29) / 40;  // :291-292
            } else {
                double sd = 0, sp = 0;
                for (int seg = 0; seg < s->fineNoncoh; ++seg) sd += cabs2(at(seg, 0, 2, kf)), sp += cabs2(at(seg, 1, 2, kf));
                v = sd + sp;  // :321
            }
            if (v > best) best = v, kbest = kf;
        }
        job0 += (size_t)(a.signal == BDS_SIGNAL_B1C ? ncomp : 2 * s->fineNoncoh) * nchunk;
        double cf = fine_frq[pi][kbest];
        if (cf == 0) cf = 1;  // :333-335
        carrFreq[prns[pi] - 1] = cf;
        codePhase[prns[pi] - 1] = (double)res[pi].codePhase;
        if (a.rs.on) {
            // results back at the original sampling rate (B2a/acquisition.m:339-356, B1C :311-328)
            codePhase[prns[pi] - 1] = std::floor((double)(res[pi].codePhase - 1) / s->samplingFreq * a.rs.old_fs) + 1;
            double doppler;
            if (s->IF >= s->samplingFreq / 2)
                doppler = (s->samplingFreq - s->IF) - cf;
            else
                doppler = cf - s->IF;
            carrFreq[prns[pi] - 1] = doppler + a.rs.old_if;
        }
        if (detected) detected[prns[pi] - 1] = 1;
    }
    return BDS_OK;
}

constexpr int kHostRefine = -1002;   // refine_device: this run needs the host path (never returned through the C ABI)
constexpr int kRefCandCap = 16384;   // candidates per stage the device chain holds (cfg3: a few hundred in the band)
constexpr int kExtra2Cap = 1 << 20;  // candidate list of the B2a second-peak pass (one cell per PRN)

bool AcqRun::device_refine_ok() const {
    const Tuning &tune = ctx->tune;
    if (!wcols || tune.neigh != 0 || tune.host_refine || a.rs.on || a.skind >= kF64 || P < 1) return false;
    if (a.signal == BDS_SIGNAL_B2A) {
        const Plan2D &pl = a.plan;
        if (!(pl.small && fsearch)) return false;  // the tile kernel's second-peak pass reports per-tile records: host path
        const size_t cap_cells = a.bw_cap / (size_t)pl.L * 8 / elem / (size_t)ncomp;
        if ((size_t)P > cap_cells) return false;
    }
    return true;
}

int AcqRun::refine_device() {
    Plan2D &pl = a.plan;
    const Tuning &tune = ctx->tune;
    const hipStream_t sm = stream();
    const bool b1c = a.signal == BDS_SIGNAL_B1C;
    int rc;
    // ---- parameters, buffers, tables ---------------------------------------------------------------
    RefParams rp{};
    rp.P = P, rp.D = D, rp.ncomp = ncomp, rp.signal = a.signal;
    rp.half = a.half && !tune.no_selfcheck ? 1 : 0;
    rp.cand_cap = kRefCandCap, rp.extra_cap = kExtraCap;
    rp.kDelta = kDelta;
    rp.f0 = f0, rp.step = s->acqStep;
    rp.X = a.X, rp.N = a.N, rp.spc = a.spc, rp.n_samples = a.n_samples;
    rp.threshold = s->acqThreshold;
    rp.s2c = (long)std::ceil(s->samplingFreq / s->codeFreqBasis) * 2;  // samples2CodeChip, B2a :137
    rp.fineNoncoh = s->fineNoncoh;
    rp.nfine = b1c ? (int)m_round(s->acqStep / 25) * 2 + 1 : (int)m_round(s->acqStep / 25) + 1;  // B1C :267, B2a :265
    rp.nchunk = (rp.nfine + kCorrFreqs - 1) / kCorrFreqs;
    rp.cplx = a.cplx ? 1 : 0;
    if (b1c) {
        if ((rc = metric_b1c_sigpower())) return rc;
        rp.sigPower = a.sigpower;
    }
    const int fine_per = (b1c ? ncomp : 2 * s->fineNoncoh) * rp.nchunk;
    if ((rc = ensure_code_cache(ctx, a, *s))) return rc;
    for (int pi = 0; pi < P; ++pi)
        for (int comp = 0; comp < ncomp; ++comp) {
            make_code_table(ctx, a, (prns[pi] - 1) * 2 + comp, 0);
            if (!b1c) make_code_table(ctx, a, (prns[pi] - 1) * 2 + comp, 1);
        }
    if ((rc = ensure_job_buffers(ctx, a, std::max<size_t>((size_t)kRefCandCap * ncomp, (size_t)P * fine_per)))) return rc;
    // everything the chain wants zeroed lives in ONE block (one fill instead of five):
    //   RefGlobal | RefPrn[P] | cellmax2[P] | lb2[P] | extra2_count
    {
        const size_t o_prn = 64, o_cm2 = o_prn + sizeof(RefPrn) * (size_t)P, o_lb2 = o_cm2 + sizeof(unsigned long long) * (size_t)P;
        const size_t o_cnt = (o_lb2 + sizeof(float) * (size_t)P + 15) & ~(size_t)15, total = o_cnt + 16;
        static_assert(sizeof(RefGlobal) <= 64 && sizeof(RefPrn) % 16 == 0, "layout of the zeroed block");
        if ((rc = ensure(ctx, &a.d_ref_zero, &a.ref_zero_cap, total))) return rc;
        a.d_ref_g = (RefGlobal *)a.d_ref_zero;
        a.d_ref_prn = (RefPrn *)(a.d_ref_zero + o_prn);
        a.d_cellmax2 = (unsigned long long *)(a.d_ref_zero + o_cm2);
        a.d_lb2 = (float *)(a.d_ref_zero + o_lb2);
        a.d_extra2_count = (int *)(a.d_ref_zero + o_cnt);
        BDS_HIP(ctx, hipMemsetAsync(a.d_ref_zero, 0, total, sm));
    }
    if ((rc = ensure(ctx, &a.d_ref_cand, &a.ref_cand_cap, (size_t)2 * kRefCandCap))) return rc;
    if ((rc = ensure(ctx, &a.d_ref_tabs, &a.ref_tabs_cap, (sizeof(long) + sizeof(int)) * (size_t)P + 64))) return rc;
    long *d_cs_of = (long *)a.d_ref_tabs;
    int *d_prn_of = (int *)(d_cs_of + P);
    std::vector<long> h_cs(P);
    for (int pi = 0; pi < P; ++pi) h_cs[pi] = (long)a.cs_slot[prns[pi]] * ncomp * pl.L;
    BDS_HIP(ctx, hipMemcpyAsync(d_cs_of, h_cs.data(), sizeof(long) * P, hipMemcpyHostToDevice, sm));
    BDS_HIP(ctx, hipMemcpyAsync(d_prn_of, prns.data(), sizeof(int) * P, hipMemcpyHostToDevice, sm));
    const unsigned pb = (unsigned)((P + 63) / 64);

    // ---- coarse refinement: thresholds -> candidates in the band -> f64 sums -> per-PRN maximum ----------------
    hipLaunchKernelGGL(k_ref_thr<false>, dim3(P), dim3(64), 0, sm, (const unsigned long long *)a.d_cellmax, rp, a.d_ref_prn, a.d_ref_g,
                       (const int *)a.d_extra_count);
    hipLaunchKernelGGL(k_ref_compact<false>, dim3(256), dim3(256), 0, sm, (const Extra *)a.d_extra, (const int *)a.d_extra_count, rp,
                       (const RefPrn *)a.d_ref_prn, (const int *)d_prn_of, (const int4 *)nullptr, a.d_ref_cand, a.d_jobs, a.d_ref_g);
    hipLaunchKernelGGL(k_corr_f64, dim3(1024, kCorrSlices), dim3(256), 0, sm, a.sview(), a.N, (const int8_t *)a.d_codes, a.code_stride,
                       1.0 / a.fs, (const CorrJob *)a.d_jobs, a.d_jobout, (const int *)&a.d_ref_g->ncand, kRefCandCap, ncomp);
    hipLaunchKernelGGL(k_ref_pick<false>, dim3(P), dim3(256), 0, sm, (const RefCand *)a.d_ref_cand, (const double2 *)a.d_jobout, kCorrSlices,
                       rp, a.d_ref_prn, a.d_ref_g);
    BDS_HIP(ctx, hipGetLastError());

    // ---- B2a: second peak of the winning bin, outside +-2 chips and within +-1 code (acquisition.m:224-249) -------
    if (!b1c) {
        if ((rc = ensure(ctx, &a.d_extra2, &a.extra2_cap, (size_t)kExtra2Cap))) return rc;
        const size_t nb_ = sizeof(int) * P + sizeof(long) * P + sizeof(int4) * P + 64;
        if ((rc = ensure(ctx, &a.d_cells, &a.cells_cap, nb_))) return rc;
        int4 *d_rng = (int4 *)a.d_cells;  // 16-byte aligned first
        long *d_cs = (long *)(d_rng + P);
        int *d_bin = (int *)(d_cs + P);
        hipLaunchKernelGGL(k_ref_second_setup, dim3(pb), dim3(64), 0, sm, rp, a.d_ref_prn, (const long *)d_cs_of, d_rng, d_cs, d_bin, a.d_ref_g);
        const SieveOut so_keep = so;
        so.recs = nullptr;
        so.extra = a.d_extra2, so.extra_count = a.d_extra2_count, so.extra_cap = kExtra2Cap;
        so.cellmax = a.d_cellmax2, so.lb = a.d_lb2, so.lb_div = 1;
        const CellList cl{d_bin, d_cs, d_rng};
        launch_list(P, nullptr, cl, 0, nullptr);
        so = so_keep;
        RefParams rp2 = rp;
        rp2.extra_cap = kExtra2Cap;
        RefCand *cand2 = a.d_ref_cand + kRefCandCap;
        hipLaunchKernelGGL(k_ref_thr<true>, dim3(P), dim3(64), 0, sm, (const unsigned long long *)a.d_cellmax2, rp2, a.d_ref_prn, a.d_ref_g,
                           (const int *)a.d_extra2_count);
        hipLaunchKernelGGL(k_ref_compact<true>, dim3(64), dim3(256), 0, sm, (const Extra *)a.d_extra2, (const int *)a.d_extra2_count, rp2,
                           (const RefPrn *)a.d_ref_prn, (const int *)d_prn_of, (const int4 *)d_rng, cand2, a.d_jobs, a.d_ref_g);
        hipLaunchKernelGGL(k_corr_f64, dim3(1024, kCorrSlices), dim3(256), 0, sm, a.sview(), a.N, (const int8_t *)a.d_codes, a.code_stride,
                           1.0 / a.fs, (const CorrJob *)a.d_jobs, a.d_jobout, (const int *)&a.d_ref_g->ncand2, kRefCandCap, ncomp);
        hipLaunchKernelGGL(k_ref_pick<true>, dim3(P), dim3(256), 0, sm, (const RefCand *)cand2, (const double2 *)a.d_jobout, kCorrSlices, rp2,
                           a.d_ref_prn, a.d_ref_g);
        BDS_HIP(ctx, hipGetLastError());
    }

    // ---- threshold + fine-Doppler search --------------------------------------------------------------------
    hipLaunchKernelGGL(k_ref_fine_jobs, dim3(P), dim3(64), 0, sm, rp, a.d_ref_prn, (const int *)d_prn_of, a.sview(), (const double *)a.d_prefix_c,
                       (const double *)a.d_prefix_cq, a.d_jobs, a.d_ref_g);
    hipLaunchKernelGGL(k_corr_f64_multi, dim3((unsigned)(P * fine_per), kCorrSlices), dim3(256), 0, sm, a.sview(), a.N, (const int8_t *)a.d_codes,
                       a.code_stride, 1.0 / a.fs, (const CorrJob *)a.d_jobs, a.d_jobout);
    const size_t pick_lds = sizeof(double) * ((size_t)(b1c ? ncomp : 2 * s->fineNoncoh) * rp.nfine + rp.nfine);
    if (pick_lds > 60000) return kHostRefine;  // (thousands of fine frequencies: the host path has no such limit)
    hipLaunchKernelGGL(k_ref_fine_pick, dim3(P), dim3(256), pick_lds, sm, rp, a.d_ref_prn, (const double2 *)a.d_jobout, kCorrSlices);
    BDS_HIP(ctx, hipGetLastError());

    // ---- the one download -------------------------------------------------------------------------------------
    std::vector<char> h_blk(64 + sizeof(RefPrn) * (size_t)P);  // RefGlobal and RefPrn[P] as they lie in the zeroed block
    std::vector<unsigned long long> h_cellmax((size_t)P * D);
    BDS_HIP(ctx, hipMemcpyAsync(h_blk.data(), a.d_ref_zero, h_blk.size(), hipMemcpyDeviceToHost, sm));
    BDS_HIP(ctx, hipMemcpyAsync(h_cellmax.data(), a.d_cellmax, sizeof(unsigned long long) * (size_t)P * D, hipMemcpyDeviceToHost, sm));
    BDS_HIP(ctx, hipStreamSynchronize(sm));
    RefGlobal h_g;
    memcpy(&h_g, h_blk.data(), sizeof(h_g));
    std::vector<RefPrn> h_prn(P);
    memcpy(h_prn.data(), h_blk.data() + 64, sizeof(RefPrn) * (size_t)P);

    // ---- the host's share: the checks of collect() / refine() in their order, then the reported numbers -------------
    a.h_rowmax.resize((size_t)P * D);
    a.h_rowarg.resize((size_t)P * D);
    for (size_t i = 0; i < h_cellmax.size(); ++i) unpack_cell(h_cellmax[i], &a.h_rowmax[i], &a.h_rowarg[i]);
    a.run_prns = prns;
    a.last.clear();
    a.last_cands.clear();
    a.cands_on_device = 0;
    n_extra = h_g.n_extra;
    a.n_extra_last = n_extra;
    const bool bad = (h_g.flags & kRefNonFinite) != 0;
    if (a.half && ((bad && !tune.no_selfcheck) || tune.test_force_fallback)) return redo(kRedoFp32, bad ? "non-finite row maximum" : "test hook");
    if (n_extra > kExtraCap && a.half) return redo(kRedoFp32, "overflow list of the sieve ran over at the fp16-storage tolerance");
    if (n_extra > kExtraCap && !a.no_fast_search) return redo(kRedoPlain, "overflow list of the sieve ran over");
    auto host_path = [&](const char *reason) {
        if (tune.verbose) fprintf(stderr, "[bds] device refinement chain hands over to the host path: %s\n", reason);
        return kHostRefine;
    };
    if (h_g.flags & kRefCandOverflow) return host_path("more candidates in the band than the chain holds");
    res.assign(P, PrnResult{});
    max_of.assign(P, 0.f);
    thr_of.assign(P, 0.f);
    for (int pi = 0; pi < P; ++pi) {
        const RefPrn &r = h_prn[pi];
        max_of[pi] = r.max_of, thr_of[pi] = r.thr;
        const double best = r.ncand > 0 ? combine(a, r.v) : -1.0;
        res[pi].peak = best;
        res[pi].fbin = r.b + 1;
        res[pi].codePhase = (long)r.lag + 1;
        if ((a.half || (fsearch && !a.no_fast_search)) && !tune.no_selfcheck && r.ncand > 0 &&
            std::fabs(best - (double)max_of[pi]) > 0.5 * kDelta * best) {
            char msg[160];
            snprintf(msg, sizeof(msg), "PRN %d: sieve maximum %.9g vs f64 %.9g (rel %.3g > %.3g)", prns[pi], (double)max_of[pi], best,
                     std::fabs(best - (double)max_of[pi]) / best, 0.5 * kDelta);
            return redo(a.half ? kRedoFp32 : kRedoPlain, msg);
        }
    }
    a.cands_on_device = std::min(h_g.ncand, kRefCandCap);
    a.cands_prns = prns;
    if (b1c) {
        if ((rc = metric_b1c())) return rc;
    } else {
        if (h_g.n_extra2 > kExtra2Cap) return host_path("candidate list of the second-peak pass ran over");  // (the host pass has the larger list)
        for (int pi = 0; pi < P; ++pi) {
            if (h_prn[pi].flags & kRefEmptyRange)
                return fail(ctx, BDS_ERR_ARG, "PRN %d: empty second-peak range (acquisition.m:248 would fail)", prns[pi]);
            res[pi].denom = h_prn[pi].nsecond > 0 ? combine(a, h_prn[pi].v2) : -1.0;
        }
    }
    for (int pi = 0; pi < P; ++pi) {
        const RefPrn &r = h_prn[pi];
        PrnResult &q = res[pi];
        const double metric = q.peak / q.denom;  // :252 / B1C :235
        const bool det = metric > s->acqThreshold;
        // (the device decided on its own evaluation of the same sums; a disagreement -- a metric within an ulp of the
        //  threshold -- or a codePhase the device adjusted differently sends the run through the host path)
        if (det != (r.detected != 0) || q.codePhase != r.codePhase) {
            if (tune.verbose)
                fprintf(stderr, "[bds] PRN %d: host metric %.17g (peak %.17g / %.17g) vs device decision %d (best %.17g second %.17g nsecond %d), codePhase %ld vs %ld\n",
                        prns[pi], metric, q.peak, q.denom, r.detected, r.best, r.second, r.nsecond, q.codePhase, r.codePhase);
            return host_path("threshold decision or code phase differ between device and host");
        }
    }
    for (int pi = 0; pi < P; ++pi) {
        const RefPrn &r = h_prn[pi];
        PrnResult &q = res[pi];
        peakMetric[prns[pi] - 1] = q.peak / q.denom;
        if (!r.detected) continue;
        q.detected = true;
        if (r.flags & kRefFineRange) {
            const long blk = b1c ? a.spc : (long)s->fineNoncoh * a.spc;
            return fail(ctx, BDS_ERR_ARG, "PRN %d: fine-search block %ld..%ld outside longSignal (%s)", prns[pi], q.codePhase,
                        q.codePhase + blk - 1, b1c ? "B1C/acquisition.m:253" : "B2a/acquisition.m:290");
        }
        const double fb = bin_freq(q.fbin - 1);
        double cf = b1c ? fb - s->acqStep + 25.0 * r.kbest : fb - s->acqStep / 2 + 25.0 * r.kbest;  // B1C :282-283, B2a :300-301
        if (cf == 0) cf = 1;  // :333-335
        carrFreq[prns[pi] - 1] = cf;
        codePhase[prns[pi] - 1] = (double)q.codePhase;
        if (detected) detected[prns[pi] - 1] = 1;
    }
    return BDS_OK;
}

int AcqRun::finish() {
    Plan2D &pl = a.plan;
    const Tuning &tune = ctx->tune;
    BDS_HIP(ctx, hipEventRecord(ev3, stream()));
    BDS_HIP(ctx, hipEventSynchronize(ev3));
    for (int pi = 0; pi < P; ++pi) a.last[prns[pi]] = res[pi];

    bds_timing &t = ctx->timing;
    memset(&t, 0, sizeof(t));
    float ms = 0;
    BDS_HIP(ctx, hipEventElapsedTime(&ms, ev0, ev3));
    t.total_ms = ms;
    BDS_HIP(ctx, hipEventElapsedTime(&ms, ev0, ev1));
    t.forward_ms = ms;
    BDS_HIP(ctx, hipEventElapsedTime(&ms, ev1, ev2));
    t.search_ms = ms;
    BDS_HIP(ctx, hipEventElapsedTime(&ms, ev2, ev3));
    t.refine_ms = ms;
    double acc = 0, acc_r = 0, acc_c = 0;
    for (int i = 0; i < nsamp; ++i) {
        BDS_HIP(ctx, hipEventElapsedTime(&ms, sa[i], sb[i]));
        acc += ms;
        if (mids) {
            BDS_HIP(ctx, hipEventElapsedTime(&ms, sa[i], sm[i]));
            acc_r += ms;
            BDS_HIP(ctx, hipEventElapsedTime(&ms, sm[i], sb[i]));
            acc_c += ms;
        }
    }
    t.rows_ms = nsamp && mids ? acc_r / nsamp : 0;
    t.cols_ms = nsamp && mids ? acc_c / nsamp : 0;
    t.n_extra = a.n_extra_last;
    t.shader_clock_GHz = 0;
    if (tune.clockprobe && pl.d_clk) {  // sampled workgroups of the wave-private passes: shader-clock over reference-clock ticks
        unsigned long long h[4];
        int wall_khz = 0;
        BDS_HIP(ctx, hipMemcpy(h, pl.d_clk, sizeof(h), hipMemcpyDeviceToHost));
        BDS_HIP(ctx, hipMemset(pl.d_clk, 0, sizeof(h)));
        BDS_HIP(ctx, hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device));
        if (h[1] + h[3]) t.shader_clock_GHz = (double)(h[0] + h[2]) / (double)(h[1] + h[3]) * wall_khz * 1e-6;
    }
    // overlapped passes: the average launch-pair duration is the search time over the pair count
    t.cell_pair_ms = nsamp ? acc / nsamp : (n_pairs_total ? t.search_ms / (double)n_pairs_total : 0);
    t.cells_per_pair = (int)cells_per_pair;
    t.n_pairs = n_pairs_total;
    t.fft_len = pl.L;
    t.n_circ = a.N;
    t.n_bins = D;
    t.n_prn = P;
    t.n_comp = ncomp;
    t.half_storage = a.half ? 1 : 0;  // 0 fp32; 1 fp16 storage (fp32 arithmetic either way)
    t.plan_l1 = pl.L1;
    t.plan_l2 = pl.L2;
    t.refine_path = dev_refined ? 1 : 0;
    {
        const bool wrows_on = fsearch && a.half && pl.L2 == 4096 && (tune.wrows != 0 || pl.small);
        t.rows_kernel = !fsearch ? 0 : wrows_on ? 2 : 1;
        t.cols_kernel = !fsearch ? 0 : pl.small ? 3 : wcols ? 2 : 1;
        const bool ilv_on = wrows_on && wcols && ncomp == 2 && ((pl.L1 == 768 && tune.ilv != 0) || pl.small);
        t.kernel_flags = (ilv_on ? 1 : 0) | (wrows_on && tune.pk != 0 ? 2 : 0);
    }
    return BDS_OK;
}

// one attempt of bds_acq_run with the storage / kernels the context is set to; kRedoFp32 / kRedoPlain: see above
int acq_run_once(bds_ctx *ctx, const bds_settings *s_in, const int32_t *prn_list, int n_prn, int max_prn, double *carrFreq,
                 double *codePhase, double *peakMetric, int32_t *detected, std::string *why) {
    if (int rc0 = check_settings(ctx, *s_in)) return rc0;  // (the resampling band edges are only visible here)
    bds_settings eff;
    const bds_settings *s = effective(s_in, &eff);
    int rc = acq_configure(ctx, *s);
    if (rc) return rc;
    AcqState &a = *ctx->acq;
    if (!a.d_sig || a.n_samples < a.N) return fail(ctx, BDS_ERR_ARG, "bds_acq_run: no IF block loaded (bds_acq_load)");
    if (a.rs.on != resample_plan(*s_in).on || (a.rs.on && a.rs.new_fs != s->samplingFreq))
        return fail(ctx, BDS_ERR_ARG, "bds_acq_run: the loaded block was conditioned for different resampling settings");
    if ((rc = bds_acq_prepare(ctx, s))) return rc;
    AcqRun r(ctx, a, s);
    if (prn_list && n_prn > 0)
        r.prns.assign(prn_list, prn_list + n_prn);
    else
        r.prns.assign(s->acqSatelliteList, s->acqSatelliteList + s->n_acq);
    int list_max = 0;
    for (int i = 0; i < s->n_acq; ++i) list_max = std::max(list_max, (int)s->acqSatelliteList[i]);
    if (max_prn < list_max) return fail(ctx, BDS_ERR_ARG, "max_prn %d < max(acqSatelliteList) %d", max_prn, list_max);
    for (int p : r.prns)
        if (!a.cs_slot.count(p)) return fail(ctx, BDS_ERR_ARG, "PRN %d of the shard is not in settings.acqSatelliteList", p);
    for (int i = 0; i < max_prn; ++i) {
        carrFreq[i] = codePhase[i] = peakMetric[i] = 0.0;  // acquisition.m:161-165
        if (detected) detected[i] = 0;
    }
    r.carrFreq = carrFreq, r.codePhase = codePhase, r.peakMetric = peakMetric, r.detected = detected;
    if ((rc = r.setup())) return rc;
    if ((rc = r.forward_all())) return rc;
    if ((rc = r.search())) return rc;
    rc = r.device_refine_ok() ? r.refine_device() : kHostRefine;
    r.dev_refined = rc == BDS_OK;
    if (rc == kHostRefine) {  // host path: the lists travel to the host, jobs are built there (rounds 1-4)
        a.cands_on_device = 0;
        if (!(rc = r.collect()) && !(rc = r.refine()) && !(rc = a.signal == BDS_SIGNAL_B1C ? r.metric_b1c() : r.second_peak_b2a()))
            rc = r.fine_search();
    }
    if (!rc) rc = r.finish();
    if (rc == kRedoFp32 || rc == kRedoPlain) *why = r.why;
    return rc;
}

}  // namespace
}  // namespace bds

extern "C" int bds_acq_run(bds_ctx *ctx, const bds_settings *s_in, const int32_t *prn_list, int n_prn, int max_prn,
                           double *carrFreq, double *codePhase, double *peakMetric, int32_t *detected) {
    if (!ctx || !s_in || !carrFreq || !codePhase || !peakMetric) return BDS_ERR_ARG;
    // at most three attempts: fp16 storage -> fp32 storage -> run-time-plan kernels
    for (int attempt = 0;; ++attempt) {
        std::string why;
        const int rc = acq_run_once(ctx, s_in, prn_list, n_prn, max_prn, carrFreq, codePhase, peakMetric, detected, &why);
        if (rc != kRedoFp32 && rc != kRedoPlain) return rc;
        if (attempt >= 2) return fail(ctx, BDS_ERR_UNSUPPORTED, "bds_acq_run: the search could not be completed (%s)", why.c_str());
        AcqState &a = *ctx->acq;
        const bool plain_kernels = rc == kRedoPlain;
        if (ctx->tune.verbose) fprintf(stderr, "[bds] search re-run (%s): %s\n", plain_kernels ? "run-time-plan kernels" : "fp32 storage", why.c_str());
        if (a.plan.small) {
            // the 80 x 4096 plan has no fp32-storage kernels of its own: re-plan (256 x 1280 for cfg2) so that the re-run takes
            // the specialised fp32 pair instead of the run-time-plan kernels on 80 x 4096
            a.no_small = true;
            a.plan.L = 0;
            bds_settings eff;
            if (int rc2 = acq_configure(ctx, *effective(s_in, &eff))) return rc2;
        }
        a.half = false;
        a.no_fast_search = a.no_fast_search || plain_kernels;
        a.sC = 1.f;
        a.cs_slot.clear();
        // (the flags stay until the configuration changes: acq_configure keeps them for the same key)
        if (int rc2 = bds_acq_prepare(ctx, s_in)) return rc2;
    }
}

#ifdef BDS_EXP_PHASES
// timing build only: the phase-clock sums of the wave-private search kernels (bds_acq_f32.h), read and cleared
extern "C" __attribute__((visibility("default"))) int bds_debug_phases(unsigned long long *out, int n) {
    unsigned long long h[128] = {};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 128; ++i) out[i] = h[i];
    unsigned long long z[128] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int bds_resample_plan(const bds_settings *s, double *new_fs, double *new_if, double *wp) {
    if (!s) return BDS_ERR_ARG;
    const ResamplePlan r = resample_plan(*s);
    if (!r.on) return 0;
    if (new_fs) *new_fs = r.new_fs;
    if (new_if) *new_if = r.new_if;
    if (wp) wp[0] = r.wp1, wp[1] = r.wp2;
    return 1;
}

extern "C" int bds_fir1_bandpass(int n_taps, double wp1, double wp2, double *b) {
    if (n_taps < 3 || !b || !(wp1 > 0 && wp1 < wp2 && wp2 < 1)) return BDS_ERR_ARG;
    const std::vector<double> h = fir1_bandpass(n_taps, wp1, wp2);
    std::copy(h.begin(), h.end(), b);
    return BDS_OK;
}

extern "C" int bds_acquire(bds_ctx *ctx, const bds_settings *s, const int8_t *samples, size_t n_samples,
                           int is_complex, int max_prn, double *carrFreq, double *codePhase, double *peakMetric,
                           int32_t *detected) {
    int rc = bds_acq_load(ctx, s, samples, n_samples, is_complex);
    if (rc) return rc;
    if ((rc = bds_acq_prepare(ctx, s))) return rc;
    return bds_acq_run(ctx, s, nullptr, 0, max_prn, carrFreq, codePhase, peakMetric, detected);
}

extern "C" int bds_acq_grid(bds_ctx *ctx, float *row_max, int32_t *row_arg, int cap) {
    if (!ctx || !ctx->acq) return BDS_ERR_ARG;
    AcqState &a = *ctx->acq;
    const int n = (int)a.h_rowmax.size();
    if (cap < n) return fail(ctx, BDS_ERR_ARG, "bds_acq_grid: capacity %d < %d", cap, n);
    for (int i = 0; i < n; ++i) {
        if (row_max) row_max[i] = a.h_rowmax[i];
        if (row_arg) row_arg[i] = a.h_rowarg[i] + 1;  // 1-based like the reference
    }
    return n;
}

extern "C" int bds_acq_candidates(bds_ctx *ctx, int prn, int32_t *bin, int64_t *lag, int cap) {
    if (!ctx || !ctx->acq) return BDS_ERR_ARG;
    if (ctx->acq->cands_on_device > 0) {  // the device refinement chain keeps its candidates on the device: fetched when asked for
        bds::AcqState &a = *ctx->acq;
        std::vector<bds::RefCand> h((size_t)a.cands_on_device);
        (void)hipSetDevice(ctx->device);
        if (hipMemcpy(h.data(), a.d_ref_cand, sizeof(bds::RefCand) * h.size(), hipMemcpyDeviceToHost) != hipSuccess)
            return bds::fail(ctx, BDS_ERR_HIP, "bds_acq_candidates: download failed");
        a.last_cands.clear();
        for (const bds::RefCand &c : h)
            if (c.pi >= 0 && c.pi < (int)a.cands_prns.size()) a.last_cands[a.cands_prns[(size_t)c.pi]].push_back({c.b, (long)c.lag});
        for (auto &kv : a.last_cands) std::sort(kv.second.begin(), kv.second.end());
        a.cands_on_device = 0;
    }
    auto it = ctx->acq->last_cands.find(prn);
    if (it == ctx->acq->last_cands.end()) return 0;
    const int n = (int)it->second.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (bin) bin[i] = it->second[(size_t)i].first + 1;   // 1-based like the reference's indices
        if (lag) lag[i] = it->second[(size_t)i].second + 1;
    }
    return n;
}

extern "C" int bds_acq_peaks(bds_ctx *ctx, int max_prn, double *peak, double *denom, int32_t *fbin) {
    if (!ctx || !ctx->acq) return BDS_ERR_ARG;
    for (int i = 0; i < max_prn; ++i) {
        if (peak) peak[i] = 0;
        if (denom) denom[i] = 0;
        if (fbin) fbin[i] = 0;
    }
    for (auto &kv : ctx->acq->last) {
        if (kv.first > max_prn) continue;
        if (peak) peak[kv.first - 1] = kv.second.peak;
        if (denom) denom[kv.first - 1] = kv.second.denom;
        if (fbin) fbin[kv.first - 1] = kv.second.fbin;
    }
    return BDS_OK;
}

BDS_DEBUG_TU_READER(bds_debug_failures_acq)
