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
#include "bds_acq_pfa.h"
#include "bds_internal.h"

namespace bds {

static const double kPi = 3.14159265358979323846;

#include "bds_acq_plan.h"  // Plan2D, choose_lengths, plan_build, the kernels' constant tables

// ---------------------------------------------------------------------------------------
// Events of a run, kept in the context across runs (a run records ~100 of them -- the timing triples of the sampled launch
// pairs -- and creating / destroying them was host time of every call): make() hands out the next one of its kind, rewind()
// starts over.
struct EventPool {
    std::vector<hipEvent_t> timed, untimed;
    size_t nt = 0, nu = 0;
    void rewind() { nt = nu = 0; }
    hipError_t make(hipEvent_t *e, unsigned flags = 0) {
        std::vector<hipEvent_t> &v = flags ? untimed : timed;
        size_t &n = flags ? nu : nt;
        if (n == v.size()) {
            hipEvent_t ne;
            const hipError_t rc_ = flags ? hipEventCreateWithFlags(&ne, flags) : hipEventCreate(&ne);
            if (rc_ != hipSuccess) return rc_;
            v.push_back(ne);
        }
        *e = v[n++];
        return hipSuccess;
    }
    ~EventPool() {
        for (hipEvent_t e : timed) (void)hipEventDestroy(e);
        for (hipEvent_t e : untimed) (void)hipEventDestroy(e);
    }
};

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
    // Host view of that block.  An int8 record is kept as the bytes it came in (interleaved I/Q for a complex one) with its
    // prefix sums at every 256th sample (round 5: the f64 copies and full-length prefix arrays of rounds 1-4 were 320 MB of host
    // writes per bds_acq_load at cfg3, 79 of the 291 ms of a cold call); the f64 block of the resampling branch keeps full arrays.
    std::vector<int8_t> h_s8;
    std::vector<double> h_cpre, h_cpre_q; // prefix sums of I (and Q) at samples 0, 256, 512, ... (exact integers below 2^53)
    std::vector<double> h_re, h_im;       // resampling branch only: the conditioned block
    std::vector<double> h_prefix;         // ... its prefix sums
    std::vector<double> h_prefix_q;       // ... of the imaginary part
    bool cplx = false;              // longSignal = I + 1i*Q (postProcessing.m:92-96)
    EventPool events;               // timing / ordering events of a run, re-used across runs
    double sample_re(long i) const { return skind >= kF64 ? h_re[(size_t)i] : (double)h_s8[(size_t)(cplx ? 2 * i : i)]; }
    double sample_im(long i) const { return !cplx ? 0.0 : skind >= kF64 ? h_im[(size_t)i] : (double)h_s8[(size_t)(2 * i + 1)]; }
    // sum of the first i samples (I part / Q part): for an int8 record the same integer, hence the same f64, in any order of adding
    double prefix(long i, int q = 0) const {
        if (skind >= kF64) return q ? h_prefix_q[(size_t)i] : h_prefix[(size_t)i];
        long acc = 0;
        const int8_t *b = h_s8.data();
        for (long m = (i >> 8) << 8; m < i; ++m) acc += cplx ? b[2 * m + q] : b[m];
        return (q ? h_cpre_q : h_cpre)[(size_t)(i >> 8)] + (double)acc;
    }
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
    double pair_gb = 0;              // bds_acq_set_pair_budget_gb (overrides the BDS_ACQ_PAIR_GB of the context's tuning)
    bool pair_gb_set = false;
    int pb_last = 0;                 // PRNs per launch pair the last run settled on, and what it was decided for: the next run with
    double pb_key[6] = {0};          // the same (P, D, L, element size, components, budget) takes it without asking the driver again
    char *d_mcells = nullptr;        // cell list of the main search of a small grid (all P x D cells in a few launch pairs) ...
    size_t mcells_cap = 0;
    std::vector<long> mcells_cs;     // ... and what it holds: uploaded again only when a run's list differs
    std::vector<int> mcells_bin;
    std::vector<long> ref_tabs_cs;   // what d_ref_tabs holds (refinement chain: code-spectrum offset and PRN of every searched PRN)
    std::vector<int> ref_tabs_prn;
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
    bool fell_back = false;          // a run of this configuration was redone with fp32 storage / on the run-time-plan kernels: the NEXT
                                     // block (bds_acq_load) starts on the default path again (round 6; until then the fallback stayed for
                                     // as long as the configuration did -- one pathological block made every later one ten times slower)
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
    // N-point plan (bds_acq_pfa.h, round 6): the cached code spectra are in its layout (53 x 12 x 3125, CRT order), the search runs its pair
    bool cs_pfa = false;
    uint4 *d_pfa_coef = nullptr;  // B fragments of the 53-point stage (pfa::make_coef_frags)
    long sigpower_X = 0;       // X the cached B1C normaliser was computed for (0: none; reset by bds_acq_load)
    double sigpower = 0;       // sqrt(var(sig(1:X)) * X), B1C/acquisition.m:150
};

void acq_state_free(AcqState *a) {
    if (!a) return;
    plan_free(a->plan);
    for (void *p : {(void *)a->d_sig, (void *)a->d_prim, (void *)a->d_Cs, (void *)a->d_Xs, (void *)a->d_Bw, (void *)a->d_pfa_coef,
                    (void *)a->d_recs, (void *)a->d_rowmax, (void *)a->d_rowarg, (void *)a->d_jobs, (void *)a->d_codes,
                    (void *)a->d_jobout, (void *)a->d_sig64, (void *)a->d_ffa, (void *)a->d_ffb, (void *)a->d_fir, (void *)a->d_cells, (void *)a->d_mcells,
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

// The inter-pass buffer also SHRINKS (round 6): a context that ran with a larger budget (serving mode: 150 GiB) and is
// then given a smaller one -- bds_acq_set_pair_budget_gb(ctx, 0), a lower BDS_ACQ_PAIR_GB -- returns the difference at its next
// search instead of holding it until bds_destroy.  Called where a run knows its final need (search()); the other ensure() sites
// only grow.  A quarter of slack (+256 MiB) is tolerated so that neighbouring settings do not reallocate back and forth.
template <class T>
static int ensure_fit(bds_ctx *ctx, T **p, size_t *cap, size_t need) {
    if (*p && *cap >= need && (double)*cap * sizeof(T) > 1.25 * (double)need * sizeof(T) + 268435456.0) {
        (void)hipFree(*p), *p = nullptr, *cap = 0;
    }
    return ensure(ctx, p, cap, need);
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
    if (fast_rows(pl.L2) && !ctx->tune.generic && !ctx->tune.generic_fwd) {
        // rows of a specialised length under a run-time-plan column pass (the 80 x 4096 plan of cfg2): the compile-time row pass
        // (first stage from global memory, last stage to the typed store; 2 080 rows of cfg2 in ~25 us against 78 on the run-time engine)
        if (a.half)
            launch_rows_fwd_any<__half2>(ctx, st(ctx), pl, nb, (const float2 *)a.d_Bw, (__half2 *)dst, dst_stride, conj_flag, scale, perm);
        else
            launch_rows_fwd_any<float2>(ctx, st(ctx), pl, nb, (const float2 *)a.d_Bw, dst, dst_stride, conj_flag, scale, 0);
    } else if (a.half)  // dst counts in stored elements (fp16 complex)
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
    const int *src = nullptr;    // 80 x 4096 plan only: the rows of listed cell g already lie at cell index src[g] of Bw -- no row pass
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
    const int want = ctx->tune.wcols_qchunk > 0 ? ctx->tune.wcols_qchunk : A.G >= 1000 ? 2 : 4;  // (see Tuning::wcols_qchunk: measured at 201 and at 6432 cells per launch)
    B.qchunk = std::max(1, std::min(want, quads));
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
            if (!cl.src) launch_rows_f<4096, NC, ST>(ctx, st_, pl, Xs, G, bin0, Cs, Bw, out_scale, cl, true);
            if (so.mid) (void)hipEventRecord(so.mid, st_);
            const SColsArgs A{(const float2 *)pl.d_tw80, pl.L2, G, Bw, pl.L, w0, w1, lo1, hi1, lo2, hi2, cl.rng, so.cellmax, so.lb, so.lb_div,
                              so.extra, so.extra_count, so.extra_cap, so.cell0, so.keep, cl.src};
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

// inter-pass work buffer in float2-sized elements per transform length: the `group` cells of one launch pair at the storage type of
// the search (fp16 complex: half an element) -- which is also what the forward pass of `group` bins needs at fp32 -- , twice that
// when the column pass runs beside the next group's row pass (BDS_ACQ_OVERLAP).  (Rounds 1-5 kept two fp32-sized halves whatever
// the mode: 20 GB at cfg3 where 5 are used -- and every GiB a context frees is a GiB the driver clears before the next user gets it.)
static size_t bw_batches(const AcqState &a, const Tuning &tune) {
    const int per_cell = a.half ? 1 : 2;                                  // float2-sized elements per cell and two components
    const int search = (a.group * a.ncomp * per_cell + 1) / 2 * (tune.overlap ? 2 : 1);
    return (size_t)std::max(std::max(search, std::min(a.group, a.D > 0 ? a.D : a.group)), 8);
}

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
    if ((rc = ensure(ctx, &a.d_sig64, &a.sig64_cap, (size_t)(sig_len + 4) * NCH))) return rc;  // (+4 samples: k_corr reads whole groups of four, bds_acq_corr.h)
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
    if (a.skind == kS8) {
        // (integers: the sequential f64 sums of rounds 1-4 were exact too, so these are the same values)
        long sa = 0, sq = 0;
        const int8_t *b = a.h_s8.data();
        for (long m = 0; m < a.N; ++m) sa += std::abs((int)b[m]), sq += (int)b[m] * (int)b[m];
        long sa2 = 0, sq2 = 0;
        for (long m = 0; m < a.n_ext - a.N; ++m) sa2 += std::abs((int)b[m]), sq2 += (int)b[m] * (int)b[m];
        a.sum_abs_ext = (double)(sa + sa2), a.sum_sq_ext = (double)(sq + sq2);
    } else {
        for (long i = 0; i < a.n_ext; ++i) {
            const long m = i < a.N ? i : i - a.N;
            const double v = a.cplx ? std::hypot(a.sample_re(m), a.sample_im(m)) : std::fabs(a.sample_re(m));
            a.sum_abs_ext += v;
            a.sum_sq_ext += v * v;
        }
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
    if (ctx->acq && ctx->acq->fell_back) {  // a new block: back to the default storage and kernels (acq_configure re-derives them)
        ctx->acq->fell_back = false;
        ctx->acq->no_small = false;
        ctx->acq->plan.L = 0;
    }
    int rc = acq_configure(ctx, *s);
    if (rc) return rc;
    AcqState &a = *ctx->acq;
    const ResamplePlan r = resample_plan(*s_in);
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nb = n_samples * (cplx ? 2 : 1);
    if ((rc = ensure(ctx, &a.d_sig, &a.sig_cap, nb + 16))) return rc;  // (+16: k_corr reads whole dwords, bds_acq_corr.h)
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
        a.h_s8.assign(samples, samples + nb);
        a.h_re.clear(), a.h_im.clear(), a.h_prefix.clear(), a.h_prefix_q.clear();
    }
    if (n_eff < a.N)
        return fail(ctx, BDS_ERR_ARG, "longSignal has %ld samples%s; acquisition needs at least %ld (acquisition.m:140)",
                    n_eff, r.on ? " after resampling" : "", a.N);
    a.n_samples = n_eff;
    a.sigpower_X = 0;
    if (r.on) {
        a.h_s8.clear(), a.h_cpre.clear(), a.h_cpre_q.clear();
        a.h_prefix.resize((size_t)n_eff + 1);
        a.h_prefix[0] = 0;
        a.h_prefix_q.clear();
        for (long i = 0; i < n_eff; ++i) a.h_prefix[(size_t)i + 1] = a.h_prefix[(size_t)i] + a.h_re[(size_t)i];
        if (cplx) {
            a.h_prefix_q.resize((size_t)n_eff + 1);
            a.h_prefix_q[0] = 0;
            for (long i = 0; i < n_eff; ++i) a.h_prefix_q[(size_t)i + 1] = a.h_prefix_q[(size_t)i] + a.h_im[(size_t)i];
        }
    } else {
        // prefix sums at every 256th sample: the host's DC means (AcqState::prefix) and the device refinement chain's (the DC of
        // the B1C fine-search block, bds_acq_refine.h) both add the few samples in between
        const size_t nc = (size_t)(n_eff >> 8) + 1;
        a.h_cpre.resize(nc);
        a.h_cpre_q.resize(cplx ? nc : 0);
        long tot = 0, tot_q = 0;
        const int8_t *b = a.h_s8.data();
        for (size_t c = 0; c < nc; ++c) {
            a.h_cpre[c] = (double)tot;
            if (cplx) a.h_cpre_q[c] = (double)tot_q;
            const long m0 = (long)c << 8, m1 = std::min(m0 + 256, n_eff);
            int s0 = 0, s1 = 0;
            if (cplx)
                for (long m = m0; m < m1; ++m) s0 += b[2 * m], s1 += b[2 * m + 1];
            else
                for (long m = m0; m < m1; ++m) s0 += b[m];
            tot += s0, tot_q += s1;
        }
        if ((rc = ensure(ctx, &a.d_prefix_c, &a.prefix_c_cap, nc))) return rc;
        BDS_HIP(ctx, hipMemcpyAsync(a.d_prefix_c, a.h_cpre.data(), sizeof(double) * nc, hipMemcpyHostToDevice, st(ctx)));
        if (cplx) {
            if ((rc = ensure(ctx, &a.d_prefix_cq, &a.prefix_cq_cap, nc))) return rc;
            BDS_HIP(ctx, hipMemcpyAsync(a.d_prefix_cq, a.h_cpre_q.data(), sizeof(double) * nc, hipMemcpyHostToDevice, st(ctx)));
        }
    }
    ext_sums(a);
    BDS_HIP(ctx, hipStreamSynchronize(st(ctx)));
    return BDS_OK;
}

// Does the N-point pair (bds_acq_pfa.h) apply to a run with these settings?  B1C with both components on N = 53 x 12 x 3125 samples
// (99.375 MS/s, acqCohT = 10), fp16 storage, no resampling, and whole spectrum bins per Doppler step: acqStep N / fs an integer
// (B1C/acquisition.m:194-198: frqBins(b) = IF - band + acqStep (b - 1), so fft(carr_b x)[k] = fft(carr_1 x)[k - (b - 1) acqStep N / fs]).
// Everything else -- and every fallback of a run (fp32 storage, run-time-plan kernels) -- takes the L-point pair.
static int pfa_shift(const bds_ctx *ctx, const AcqState &a, const bds_settings &s) {
    if (!ctx->tune.pfa || !a.half || a.no_fast_search || a.signal != BDS_SIGNAL_B1C || a.ncomp != 2 || a.N != pfa::NP || a.rs.on) return 0;
    const double sh = s.acqStep * (double)a.N / s.samplingFreq;
    const long D = (long)m_round(s.acqSearchBand * 2 / s.acqStep) + 1;
    if (!(sh >= 1.0) || sh != std::floor(sh) || sh * (double)D >= (double)a.N) return 0;
    return (int)sh;
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
    if ((rc = ensure(ctx, &a.d_Bw, &a.bw_cap, bw_batches(a, ctx->tune) * (size_t)pl.L))) return rc;
    {   // layout of the cached code spectra: the N-point plan's or the L-point plan's; a change drops the cache
        const bool want = pfa_shift(ctx, a, *s) > 0;
        if (want != a.cs_pfa) a.cs_slot.clear();
        a.cs_pfa = want;
        if (a.half) a.sC = (float)std::exp2(std::floor(std::log2(32768.0 * (double)(want ? a.N : pl.L) / (double)a.X)));
        if (want && !a.d_pfa_coef) {
            std::vector<uint16_t> cf(pfa::kCoefBytes / 2);
            pfa::make_coef_frags(cf.data());
            BDS_HIP(ctx, hipMalloc((void **)&a.d_pfa_coef, pfa::kCoefBytes));
            BDS_HIP(ctx, hipMemcpy(a.d_pfa_coef, cf.data(), pfa::kCoefBytes, hipMemcpyHostToDevice));
        }
    }
    std::vector<int> todo;
    for (int i = 0; i < s->n_acq; ++i)
        if (!a.cs_slot.count(s->acqSatelliteList[i]) &&
            std::find(todo.begin(), todo.end(), s->acqSatelliteList[i]) == todo.end())
            todo.push_back(s->acqSatelliteList[i]);
    if (todo.empty()) return BDS_OK;
    RoctxRange rg("acq.prepare (code spectra)");
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
    const int chunk = (int)bw_batches(a, ctx->tune);
    for (int prn : todo) {
        const int slot = (int)a.cs_slot.size();
        CodeLoader ld{a.tab, (prn - 1) * 2};
        // components of one PRN are adjacent code slots: batch index = component
        (void)chunk;
        // half storage: the same byte buffer holds 4-byte elements, so offsets count in those
        float2 *cs_dst = a.half ? (float2 *)((__half2 *)a.d_Cs + (size_t)slot * a.ncomp * pl.L)
                                : a.d_Cs + (size_t)slot * a.ncomp * pl.L;
        if (a.cs_pfa) {  // conj(fft(code)) / N in the CRT layout, [slot][component][53][12][3125] (the slots keep the L-point stride)
            pfa::forward(st(ctx), ld, a.ncomp, a.d_Bw, (uint32_t *)cs_dst, pfa::NP, 1, (float)((double)a.sC / (double)a.N), 0);
            BDS_HIP(ctx, hipGetLastError());
        } else if ((rc = forward(ctx, a, ld, a.ncomp, cs_dst, pl.L, 1, (float)((double)a.sC / (double)pl.L))))
            return rc;
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
        hipError_t e = hipMalloc((void **)&a.d_codes, ntab * (size_t)stride + 16);  // (+16: k_corr reads whole dwords)
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

// multi: every job carries up to kCorrFreqs frequencies; out[j * kCorrFreqs + f]
// nc: jobs come in groups of nc that differ only in their code slot (the components of one candidate / segment) and are
// summed in one pass (bds_acq_corr.h)
static int run_jobs(bds_ctx *ctx, AcqState &a, const bds_settings &s, std::vector<CorrJob> &jobs,
                    std::vector<double2> &out, int nc, bool multi = false) {
    const int nper = multi ? kCorrFreqs : 1;
    out.resize(jobs.size() * nper);
    if (jobs.empty()) return BDS_OK;
    if (nc < 1 || nc > 2 || jobs.size() % (size_t)nc) return fail(ctx, BDS_ERR_ARG, "run_jobs: %zu jobs in groups of %d", jobs.size(), nc);
    int rc;
    // sampled codes the jobs refer to (built once per (slot, mode), cached in the context)
    if ((rc = ensure_code_cache(ctx, a, s))) return rc;
    for (const CorrJob &j : jobs) make_code_table(ctx, a, j.slot, j.mode);
    constexpr int kSlices = kCorrSlices;
    if ((rc = ensure_job_buffers(ctx, a, jobs.size()))) return rc;
    BDS_HIP(ctx, hipMemcpyAsync(a.d_jobs, jobs.data(), sizeof(CorrJob) * jobs.size(), hipMemcpyHostToDevice, st(ctx)));
    const dim3 grid((unsigned)(jobs.size() / (size_t)nc), kSlices);
    if (multi)
        launch_corr<kCorrFreqs>(st(ctx), grid, a.sview(), nc, a.N, (const int8_t *)a.d_codes, a.code_stride, 1.0 / a.fs, (const CorrJob *)a.d_jobs,
                                a.d_jobout, nullptr, 0);
    else
        launch_corr<1>(st(ctx), grid, a.sview(), nc, a.N, (const int8_t *)a.d_codes, a.code_stride, 1.0 / a.fs, (const CorrJob *)a.d_jobs,
                       a.d_jobout, nullptr, 0);
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
    int pfa = 0;               // > 0: the N-point pair runs (bds_acq_pfa.h); the value = spectrum bins per Doppler step
    size_t cell_elems = 0;     // fp16-complex-sized elements of one cell in the inter-pass buffer (both components)
    bool dev_refined = false;  // the refinement ran as the device chain
    size_t elem = 8;  // bytes of one stored complex value
    int PB = 1;
    long n_pairs_total = 0, cells_per_pair = 0;
    SieveOut so{};
    // timing
    EventPool &evp;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    hipEvent_t sa[kSamples], sb[kSamples], sm[kSamples];
    hipEvent_t ev_rows[2] = {nullptr, nullptr}, ev_cols[2] = {nullptr, nullptr};
    int nsamp = 0;
    double samp_cells = 0;  // cells of the sampled multi-PRN pairs
    bool mids = false;  // the sampled pairs carry a mid event (fp32-arithmetic kernels)
    size_t half_bytes = 0;  // one group of cells in the inter-pass buffer
    // the sieve's output and the decisions made on it
    int n_extra = 0;
    std::vector<Extra> h_extra;
    std::vector<float> thr_of, max_of;
    std::vector<std::vector<Cell>> cells;
    std::vector<PrnResult> res;

    AcqRun(bds_ctx *c, AcqState &st_, const bds_settings *s_) : ctx(c), a(st_), s(s_), evp(st_.events) { evp.rewind(); }
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
    pfa = a.cs_pfa ? pfa_shift(ctx, a, *s) : 0;  // (bds_acq_prepare laid the code spectra out for it with these very settings)
    if (a.cs_pfa && !pfa) return fail(ctx, BDS_ERR_HIP, "internal: code spectra in the N-point layout for a run that cannot use them");
    if ((rc = ensure(ctx, &a.d_Bw, &a.bw_cap, bw_batches(a, ctx->tune) * (size_t)pl.L))) return rc;
    // (N-point pair: ONE signal spectrum, every row stored twice -- 2 N fp16 complex = N float2-sized elements)
    if ((rc = ensure(ctx, &a.d_Xs, &a.xs_cap, pfa ? (size_t)pfa::NP : (size_t)D * pl.L))) return rc;

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
        // (N-point pair: transform length N, rows of 3125 points; its row pass has no output scale -- sB rides on the stored signal
        //  spectrum, whose rms stays ~0.2 whatever the block: forward_all)
        const double b_rms = pfa ? std::sqrt(a.sum_sq_ext) * std::sqrt((double)a.X) / (double)a.N * std::sqrt((double)pfa::K3)
                                 : std::sqrt(a.sum_sq_ext) * std::sqrt((double)a.X) / (double)pl.L * std::sqrt((double)pl.L2);
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
    fsearch = (pl.fast || (pl.small && a.half) || pfa) && !a.no_fast_search;  // fp32-arithmetic specialised kernels (default)
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
    wcols = fsearch && (pfa || pl.small || (tune.wcols != 0 && (pl.L1 != 256 || tune.wcols > 0)));
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
    cell_elems = pfa ? pfa::kCellElems : (size_t)ncomp * pl.L;  // stored complex values of one cell in the inter-pass buffer
    // One launch pair carries the whole Doppler rows of SEVERAL PRNs through a cell list: the grids fill the chip, a row workgroup
    // walks all the bins of one PRN (its code rows and twiddles set up once per D cells), the row workgroups of different PRNs
    // read the same spectrum rows at about the same time, and a call is a few long launches instead of many short ones.
    //   cfg2 (63 PRNs x 26 bins, round 3): 104 cells per pair 3.9 ms, 208 -> 3.6, 416 -> 3.2, 832 -> 3.1, all 1638 -> 3.0.
    //   cfg3 (63 PRNs x 201 bins, round 5, profiles/r05_multiprn_full*.txt, per call on one box): one PRN per pair 197.0 ms,
    //   8 PRNs 197.5, 11 PRNs 199.1 (202.3), 16 PRNs 192.8, 21 PRNs 191.8 (196.7), 32 PRNs 195.0 (202.3) -- 2.5 - 3.6 % for an
    //   inter-pass buffer of 106 - 162 GB, which is what 288 GB of HBM are for.
    // What that costs is memory, and time when it changes hands: a FRESH hipMalloc of 150 GiB takes 0.4 ms on these boxes, but freed
    // device memory is cleared by the driver at ~33 GB/s and an allocation that needs it waits -- 150 GiB allocated again after a free:
    // 4.5 - 4.8 s (tools/probe/alloc_probe.hip, profiles/r05_alloc_probe.txt).  A first call in a fresh process costs the same in both
    // modes (bench.py `cold`: 234 vs 266 ms); a process that builds and destroys contexts, or the next process on the device, pays for
    // the clearing.  So the budget is the deployment's to choose (BDS_ACQ_PAIR_GB / bds_acq_set_pair_budget_gb): the DEFAULT is
    // 40 GiB -- 8 PRNs per pair at cfg3, 2.5 % less time per call than one PRN per pair, a footprint a host can plan with (rounds
    // 1-5 allocated 20 GB whatever the mode) --, "auto" = 60 % of the device memory that is free (counting what this context already
    // holds: 32 + 31 PRNs, 150 GiB, another 2.5 %) is the SERVING mode of a process that keeps the device to itself, 0 the minimal
    // footprint (one PRN's row per pair, 5 GB at cfg3).  With the default budget a big grid is batched only when at least six PRNs
    // fit a pair (fewer gain nothing: 4 PRNs 197.8 against 196.7 ms); an explicit budget batches from two.
    // Small grids (D <= 104: cfg2's 63 x 26 cells are 4.3 GB) batch up to the budget too (8 GiB when the budget is 0).  The PRNs are dealt evenly over the
    // pairs; with room for less than two PRNs' cells a pair is one group of one PRN.
    // (BDS_ACQ_NOMULTI / BDS_ACQ_MULTI_ANY / BDS_ACQ_PBCELLS / BDS_ACQ_PBCAP_GB of the hooks build: off / on / cells per pair / budget.)
    const double pair_gb = tune.pbcap_gb > 0 ? tune.pbcap_gb : a.pair_gb_set ? a.pair_gb : tune.pair_gb;
    multiprn = fsearch && P > 1 && !tune.nomulti && (D <= 104 || tune.multi_any || pair_gb != 0);
    PB = 1;
    const double pb_key[6] = {(double)P, (double)D, (double)cell_elems, (double)elem, (double)ncomp, pair_gb + (tune.pbcells ? 1e6 * tune.pbcells : 0)};
    const bool pb_known = multiprn && a.pb_last > 0 && !memcmp(pb_key, a.pb_key, sizeof(pb_key));
    if (pb_known) {
        PB = a.pb_last;
        multiprn = PB >= 2;
    } else if (multiprn) {
        double budget = (pair_gb > 0 ? pair_gb : 8.0) * 1073741824.0;
        if (pair_gb < 0) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = (size_t)16 << 30;
            budget = 0.6 * ((double)fr + (double)a.bw_cap * sizeof(float2));
        }
        const long pb_cap = (long)(budget / ((double)cell_elems * (double)elem));  // cells
        const long pb_cells = tune.pbcells ? std::min<long>(tune.pbcells, pb_cap) : pb_cap;
        const long pb_max = std::min<long>(P, std::max<long>(D <= 104 ? 2 : 1, pb_cells / D));
        const bool explicit_budget = tune.pbcap_gb > 0 || a.pair_gb_set || tune.pair_gb_env || tune.pbcells > 0 || tune.multi_any;
        if (pb_max < (D <= 104 || explicit_budget ? 2 : 6)) {
            multiprn = false;
        } else {
            const long np_ = (P + pb_max - 1) / pb_max;
            PB = (int)((P + np_ - 1) / np_);
        }
    }
    if (!pb_known && !tune.nomulti && fsearch && P > 1) {
        memcpy(a.pb_key, pb_key, sizeof(pb_key));
        a.pb_last = multiprn ? PB : 1;
    }
    n_pairs_total = (long)P * ((D + G - 1) / G);
    cells_per_pair = G;
    if (multiprn) n_pairs_total = (P + PB - 1) / PB, cells_per_pair = (long)PB * D;
    if (pfa && !multiprn) multiprn = true, PB = 1, n_pairs_total = P, cells_per_pair = D;  // the N-point pair always runs on cell lists
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
    if (pfa) {  // ONE transform: the spectrum of bin 0; bin b is its rotation by b * pfa bins (bds_acq_pfa.h)
        SignalLoader ld{a.sview(), a.N, a.n_ext, f0, s->acqStep, 1.0 / a.fs, 0};
        pfa::forward(stream(), ld, 1, a.d_Bw, (uint32_t *)a.d_Xs, 0, 0, a.sX * a.sB, 1);
        BDS_HIP(ctx, hipGetLastError());
        BDS_HIP(ctx, hipEventRecord(ev1, stream()));
        return BDS_OK;
    }
    const int chunk = (int)bw_batches(a, ctx->tune);
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
    if (pfa) {  // the N-point pair (bds_acq_pfa.h): every lag of the N is searched, the cells come as a list
        const int gc = std::max(1, cl.gc), chunks = (ncells + gc - 1) / gc;
        const size_t rows_lds = 2 * 3136 * sizeof(float2);
        want_lds(ctx, pfa::k_pfa_cols<2, false>, pfa::kColsLds);
        pfa::RowsArgs ra{(const uint32_t *)a.d_Xs, (const uint32_t *)a.d_Cs, (uint32_t *)a.d_Bw, cl.bin, cl.cs, ncells, gc, pfa};
        hipLaunchKernelGGL(pfa::k_pfa_rows<2>, dim3((unsigned)(pfa::MP * pfa::K2 * chunks)), dim3(pfa::kRowsThreads), rows_lds, s_main, ra);
        if (mid) (void)hipEventRecord(mid, s_main);
        const int qch = ctx->tune.pfa_qchunk > 0 ? ctx->tune.pfa_qchunk : 1;  // (one tile of a cell, then the same tile of the next cell)
        const long items = (long)((pfa::kTiles + qch - 1) / qch) * qch * ncells;
        // (a workgroup loads the 57 KB coefficient table once: at least ~24 items each when the launch is small -- one PRN's row per pair)
        const unsigned cgrid = (unsigned)std::min<long>(items, ctx->tune.pfa_cgrid > 0 ? ctx->tune.pfa_cgrid : std::max<long>(512, std::min<long>(8192, items / 24)));
        pfa::ColsArgs ca{(const uint32_t *)a.d_Bw, a.d_pfa_coef, ncells, w0, w1, so1.cellmax, so1.lb, so1.lb_div, so1.extra, so1.extra_count,
                         so1.extra_cap, cell0, so1.keep, qch, nullptr, nullptr, -1, -1};
        hipLaunchKernelGGL((pfa::k_pfa_cols<2, false>), dim3(cgrid), dim3(pfa::kColsThreads), pfa::kColsLds, s_main, ca);
        return;
    }
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
    if (!multiprn && (rc = ensure_fit(ctx, &a.d_Bw, &a.bw_cap, bw_batches(a, ctx->tune) * (size_t)pl.L))) return rc;  // (gives a larger budget's buffer back)
    if (multiprn) {
        // float2-sized elements the PB*D cells of one launch pair occupy; if the device cannot give that much after all (another
        // process took it since setup() asked), halve the PRNs per pair
        for (;;) {
            const size_t need = ((size_t)PB * D * cell_elems * elem + 7) / 8;
            if (!(rc = ensure_fit(ctx, &a.d_Bw, &a.bw_cap, std::max(need, bw_batches(a, ctx->tune) * (size_t)pl.L)))) break;
            if (PB <= 2) return rc;
            (void)hipGetLastError();  // (the failed hipMalloc is handled here: it must not surface at the end of the search)
            PB = (PB + 1) / 2;
            a.pb_last = PB;
            n_pairs_total = (P + PB - 1) / PB, cells_per_pair = (long)PB * D;
            if (ctx->tune.verbose) fprintf(stderr, "[bds] inter-pass buffer: allocation failed, %d PRNs per launch pair instead\n", PB);
        }
        const size_t nc_ = (size_t)P * D;
        std::vector<int> h_bin(nc_);
        std::vector<long> h_cs(nc_);
        for (int pi = 0; pi < P; ++pi)
            for (int b = 0; b < D; ++b) {
                h_bin[(size_t)pi * D + b] = b;
                h_cs[(size_t)pi * D + b] = (long)a.cs_slot[prns[pi]] * ncomp * pl.L;
            }
        const char *before = a.d_mcells;
        if ((rc = ensure(ctx, &a.d_mcells, &a.mcells_cap, (sizeof(long) + sizeof(int)) * nc_ + 64))) return rc;
        long *d_cs = (long *)a.d_mcells;
        int *d_bin = (int *)(d_cs + nc_);
        if (a.d_mcells != before || a.mcells_cs != h_cs || a.mcells_bin != h_bin) {  // (the same PRN list call after call: no upload)
            a.mcells_cs.clear(), a.mcells_bin.clear();
            BDS_HIP(ctx, hipMemcpyAsync(d_cs, h_cs.data(), sizeof(long) * nc_, hipMemcpyHostToDevice, s_main));
            BDS_HIP(ctx, hipMemcpyAsync(d_bin, h_bin.data(), sizeof(int) * nc_, hipMemcpyHostToDevice, s_main));
            BDS_HIP(ctx, hipStreamSynchronize(s_main));  // (pageable sources: the copies are done before the vectors go)
            a.mcells_cs = h_cs, a.mcells_bin = h_bin;
        }
        for (int pi0 = 0; pi0 < P; pi0 += PB, ++pair_idx) {
            const int np_ = std::min(PB, P - pi0);
            CellList cl;
            cl.bin = d_bin + (size_t)pi0 * D;
            cl.cs = d_cs + (size_t)pi0 * D;
            // Cells one row workgroup walks: all D bins of its PRN if the launch then still has >= 32 waves of row workgroups for the
            // chip's 512 slots (their tail is what a short launch of long workgroups loses: 8 PRNs per pair in 201-bin workgroups
            // 197.2-197.5 ms per cfg3 call = the lean mode's 197.0, in 67-bin chunks 192.0-192.3; 32 PRNs the same either way,
            // profiles/r05_small_pairs*.txt), else the largest divisor of D that gives them, but no chunk below 32 cells (a
            // workgroup's set-up: code rows, 60 twiddle registers).  A chunk stays inside one PRN's cells, hence a divisor.
            cl.gc = D;
            if (ctx->tune.list_gc > 0 && D % ctx->tune.list_gc == 0) {
                cl.gc = ctx->tune.list_gc;
            } else if (ctx->tune.list_gc == 0) {
                const long want_wgs = 32L * 512;
                for (int dv = 1; dv <= D; ++dv) {
                    if (D % dv) continue;
                    const int gc_ = D / dv;  // dv chunks per PRN and row
                    if (gc_ < 32 && dv > 1) break;
                    cl.gc = gc_;
                    if ((long)(pfa ? pfa::MP * pfa::K2 : pl.L1) * np_ * dv >= want_wgs) break;
                }
            }
            // (a call is a handful of pairs: all of them are timed, the last, shorter one included -- cell_pair_ms and
            //  cells_per_pair are then the means a kernel trace of the call shows)
            const bool sample = (np_ == PB || n_pairs_total <= kSamples) && (pair_idx % sample_every) == 0 && nsamp < kSamples;
            if (sample) BDS_HIP(ctx, hipEventRecord(sa[nsamp], s_main));
            launch_list(np_ * D, wcols ? nullptr : a.d_recs + (size_t)pi0 * D * pl.ntiles, cl, pi0 * D, sample ? sm[nsamp] : nullptr);
            if (sample) {
                BDS_HIP(ctx, hipEventRecord(sb[nsamp++], s_main));
                samp_cells += (double)np_ * D;
            }
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

#include "bds_acq_decide.h"  // AcqRun::collect .. fine_search (host path), AcqRun::refine_device (device chain)

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
    t.cells_per_pair = samp_cells > 0 && nsamp ? samp_cells / nsamp : (double)cells_per_pair;
    t.n_pairs = n_pairs_total;
    t.fft_len = pfa ? pfa::NP : pl.L;
    t.n_circ = a.N;
    t.n_bins = D;
    t.n_prn = P;
    t.n_comp = ncomp;
    t.half_storage = a.half ? 1 : 0;  // 0 fp32; 1 fp16 storage (fp32 arithmetic either way)
    t.plan_l1 = pfa ? pfa::K1 * pfa::K2 : pl.L1;
    t.plan_l2 = pfa ? pfa::K3 : pl.L2;
    t.refine_path = dev_refined ? 1 : 0;
    {
        const bool wrows_on = fsearch && a.half && pl.L2 == 4096 && (tune.wrows != 0 || pl.small);
        t.rows_kernel = !fsearch ? 0 : pfa ? 3 : wrows_on ? 2 : 1;
        t.cols_kernel = !fsearch ? 0 : pfa ? 4 : pl.small ? 3 : wcols ? 2 : 1;
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
    RoctxRange whole("bds_acq_run");
    if ((rc = r.setup())) return rc;
    {
        RoctxRange rg("acq.forward");
        if ((rc = r.forward_all())) return rc;
    }
    {
        RoctxRange rg("acq.search");
        if ((rc = r.search())) return rc;
    }
    {
        RoctxRange rg("acq.refine");
        rc = r.device_refine_ok() ? r.refine_device() : kHostRefine;
        r.dev_refined = rc == BDS_OK;
        if (rc == kHostRefine) {  // host path: the lists travel to the host, jobs are built there (rounds 1-4)
            a.cands_on_device = 0;
            if (!(rc = r.collect()) && !(rc = r.refine()) && !(rc = a.signal == BDS_SIGNAL_B1C ? r.metric_b1c() : r.second_peak_b2a()))
                rc = r.fine_search();
        }
        if (!rc) rc = r.finish();
    }
    if (rc == kRedoFp32 || rc == kRedoPlain) *why = r.why;
    return rc;
}

}  // namespace
}  // namespace bds

/* Serving mode of the search (see AcqRun::setup): budget of the inter-pass buffer in GiB; 0 = the minimal footprint (one PRN per pair), < 0 = 60 % of the free device memory; unset: Tuning::pair_gb (40) */
extern "C" int bds_acq_set_pair_budget_gb(bds_ctx *ctx, double gib) {
    if (!ctx) return BDS_ERR_ARG;
    if (!(gib == gib)) return fail(ctx, BDS_ERR_ARG, "bds_acq_set_pair_budget_gb: not a number");
    if (!ctx->acq) ctx->acq = new AcqState();
    ctx->acq->pair_gb = gib;
    ctx->acq->pair_gb_set = true;
    return BDS_OK;
}

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
        a.fell_back = true;
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

extern "C" int bds_acq_coherent_sums(bds_ctx *ctx, const bds_settings *s_in, int prn, int64_t phase, const double *freqs, int nf, int mode,
                                     double *out, int cap) {
    using namespace bds;
    if (!ctx || !ctx->acq || !s_in || !freqs || !out || nf < 1 || nf > 4096 || mode < 0 || mode > 2 || prn < 1 || prn > BDS_MAX_PRN || cap < 0)
        return BDS_ERR_ARG;
    AcqState &a = *ctx->acq;
    bds_settings eff;
    const bds_settings *s = effective(s_in, &eff);
    if (a.n_samples <= 0 || a.spc <= 0) return fail(ctx, BDS_ERR_ARG, "bds_acq_coherent_sums: no block loaded (bds_acq_load / bds_acq_run first)");
    (void)hipSetDevice(ctx->device);
    const bool b1c = a.signal == BDS_SIGNAL_B1C;
    const int nc = b1c ? a.ncomp : 2;
    {  // (re, im) pairs the call writes: nothing is written when `out` cannot hold them (round 6: the entry had no capacity argument)
        const long need = mode == 0 ? (long)nf * a.ncomp : (long)(b1c ? 1 : s->fineNoncoh) * nc * nf;
        if (need > cap) return fail(ctx, BDS_ERR_ARG, "bds_acq_coherent_sums: out holds %d (re, im) pairs, the call writes %ld", cap, need);
    }
    std::vector<CorrJob> jobs;
    std::vector<double2> jout;
    int rc;
    if (mode == 0) {
        // the coarse cell (frequency, code phase): acquisition.m:194-209 -- circular, X samples, the sampled code table
        if (phase < 1 || phase > a.N) return BDS_ERR_ARG;
        for (int f = 0; f < nf; ++f)
            for (int c = 0; c < a.ncomp; ++c) {
                CorrJob j{};
                j.start = phase - 1, j.len = a.X, j.freq = freqs[f], j.slot = (prn - 1) * 2 + c, j.circ = 1, j.mode = 0;
                jobs.push_back(j);
            }
        if ((rc = run_jobs(ctx, a, *s, jobs, jout, a.ncomp))) return rc;
        for (size_t i = 0; i < jout.size(); ++i) out[2 * i] = jout[i].x, out[2 * i + 1] = jout[i].y;
        return (int)jout.size();
    }
    // the fine-search block starting at code phase `phase` (B2a/acquisition.m:287-316, B1C/acquisition.m:253-287)
    const int nseg = b1c ? 1 : s->fineNoncoh;
    const long blk = (long)nseg * a.spc;
    if (phase < 1 || phase - 1 + blk > a.n_samples) return BDS_ERR_ARG;
    double mean = 0, mean_q = 0;
    if (b1c) {
        mean = (a.prefix(phase - 1 + a.spc) - a.prefix(phase - 1)) / (double)a.spc;
        mean_q = a.cplx ? (a.prefix(phase - 1 + a.spc, 1) - a.prefix(phase - 1, 1)) / (double)a.spc : 0.0;
    }
    const bool multi = mode == 1;
    const int per = multi ? kCorrFreqs : 1, nchunk = (nf + per - 1) / per;
    for (int seg = 0; seg < nseg; ++seg)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int c = 0; c < nc; ++c) {
                CorrJob j{};
                j.start = phase - 1 + (long)seg * a.spc, j.len = a.spc, j.code_k0 = b1c ? 0 : (long)seg * a.spc;
                j.mean = mean, j.mean_q = mean_q, j.slot = (prn - 1) * 2 + c, j.circ = 0, j.mode = b1c ? 0 : 1;
                j.nf = std::min(per, nf - ch * per);
                for (int f = 0; f < j.nf; ++f) j.fr[f] = freqs[ch * per + f];
                j.freq = j.fr[0];
                jobs.push_back(j);
            }
    if ((rc = run_jobs(ctx, a, *s, jobs, jout, nc, multi))) return rc;
    // out[((seg * nc + c) * nf + f) * 2 + {re, im}]
    for (int seg = 0; seg < nseg; ++seg)
        for (int c = 0; c < nc; ++c)
            for (int f = 0; f < nf; ++f) {
                const size_t j = ((size_t)seg * nchunk + f / per) * nc + c;
                const double2 v = jout[j * per + f % per];
                out[(((size_t)seg * nc + c) * nf + f) * 2] = v.x, out[(((size_t)seg * nc + c) * nf + f) * 2 + 1] = v.y;
            }
    return nseg * nc * nf;
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
