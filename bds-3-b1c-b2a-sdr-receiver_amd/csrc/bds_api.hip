// Context management, error reporting and the small host-side helpers of the C ABI.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <numeric>

#include "bds_internal.h"

namespace bds {

static thread_local std::string g_create_error;

int fail(bds_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

// Environment knobs.  The RELEASE library reads four documented ones (include/bds_mi355x.h lists them):
//   BDS_ACQ_FP16=0      fp32 storage of the spectra and the inter-pass buffer as well (default: fp16 storage, f64 decisions)
//   BDS_TRK_PREC=0..5   numerics of the tracking correlator (default 4 = strict, a sin / cos of the reference's own carrier argument per
//                       sample; 5 = the same argument by angle addition, 12 % faster, 1e-10 instead of 1e-13 from the oracle; 0 = fp32 carrier)
//   BDS_VERBOSE         progress / fallback messages on stderr
//   BDS_ACQ_CLOCKPROBE  sampled workgroups time themselves with the shader clock (bds_timing::shader_clock_GHz)
//   BDS_ACQ_PAIR_GB     budget of the search's inter-pass buffer in GiB: a launch pair carries as many PRNs' Doppler rows as fit (default
//                       40; "auto": 60 % of the free device memory, the serving mode; 0: minimal, one PRN per pair); bds_acq_set_pair_budget_gb
//                       is the same switch for a host program
// Everything else -- kernel selection, launch shapes, plan overrides, the sieve tolerance, the switches that turn the
// completeness self-check off or force a fallback -- exists only in the TEST-HOOKS build (BDS_TEST_HOOKS=1 ./build.sh ->
// libbds_mi355x_hooks.so, what tests/ load): a stray variable in a MATLAB session cannot change what the release library decides.
// ---- ROCTx stage markers (bds_internal.h) --------------------------------------------------------------------------------
namespace {
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)();
struct Roctx {
    roctx_push_fn push = nullptr;
    roctx_pop_fn pop = nullptr;
    Roctx() {
        for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void *lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!lib) continue;
            push = (roctx_push_fn)dlsym(lib, "roctxRangePushA");
            pop = (roctx_pop_fn)dlsym(lib, "roctxRangePop");
            if (push && pop) return;
            push = nullptr, pop = nullptr;
        }
    }
};
const Roctx &roctx() {
    static const Roctx r;  // (thread-safe initialisation)
    return r;
}
}  // namespace
void roctx_push(const char *name) {
    if (roctx().push) roctx().push(name);
}
void roctx_pop() {
    if (roctx().pop) roctx().pop();
}

Tuning tuning_from_env() {
    Tuning t;
    auto geti = [](const char *name, int dflt) {
        const char *e = std::getenv(name);
        return e ? std::atoi(e) : dflt;
    };
    auto has = [](const char *name) { return std::getenv(name) != nullptr; };
    t.fp16_storage = geti("BDS_ACQ_FP16", -1);
    t.trk_prec = geti("BDS_TRK_PREC", 4);
    t.verbose = has("BDS_VERBOSE");
    t.clockprobe = geti("BDS_ACQ_CLOCKPROBE", 0);
    if (const char *e = std::getenv("BDS_ACQ_PAIR_GB")) t.pair_gb = (e[0] == 'a' || e[0] == 'A') ? -1.0 : std::atof(e), t.pair_gb_env = true;
#ifdef BDS_TEST_HOOKS
    if (const char *e = std::getenv("BDS_ACQ_FORCE_L1L2")) {
        int a = 0, b = 0;
        if (sscanf(e, "%dx%d", &a, &b) == 2) t.force_l1 = a, t.force_l2 = b;
    }
    t.logt = geti("BDS_ACQ_LOGT", -1);
    t.generic = has("BDS_ACQ_GENERIC");
    t.generic_fwd = has("BDS_ACQ_GENERIC_FWD");
    t.group = std::max(0, std::min(1024, geti("BDS_ACQ_GROUP", 0)));
    t.gchunk = std::max(1, geti("BDS_ACQ_GCHUNK", 34));
    t.multi_any = has("BDS_ACQ_MULTI_ANY");
    t.nomulti = has("BDS_ACQ_NOMULTI");
    t.pbcells = std::max(0, geti("BDS_ACQ_PBCELLS", 0));
    if (const char *e = std::getenv("BDS_ACQ_PBCAP_GB")) t.pbcap_gb = std::atof(e);
    t.rows_grid = std::max(0, geti("BDS_ACQ_ROWS_GRID", 0));
    t.overlap = has("BDS_ACQ_OVERLAP");
    t.no_bwreuse = has("BDS_ACQ_NO_BWREUSE");
    t.list_gc = std::max(-1, geti("BDS_ACQ_LIST_GC", 0));
    if (const char *e = std::getenv("BDS_ACQ_KDELTA")) t.kdelta = std::max(0.0, std::min(0.9, std::atof(e)));
    t.no_selfcheck = has("BDS_ACQ_NO_SELFCHECK");
    t.test_force_fallback = has("BDS_ACQ_TEST_FORCE_FALLBACK");
    t.wcols = geti("BDS_ACQ_WCOLS", -1);
    t.wrows = geti("BDS_ACQ_WROWS", -1);
    t.ilv = geti("BDS_ACQ_ILV", 1);
    t.pk = geti("BDS_ACQ_PK", 1);
    t.small_plan = geti("BDS_ACQ_SMALL", 1);
    t.pfa = geti("BDS_ACQ_PFA", 1);
    t.pfa_qchunk = std::max(0, geti("BDS_ACQ_PFA_QCHUNK", 0));
    t.pfa_cgrid = std::max(0, geti("BDS_ACQ_PFA_CGRID", 0));
    t.host_refine = has("BDS_ACQ_HOSTREFINE");
    t.neigh = std::max(0, std::min(4, geti("BDS_ACQ_NEIGH", 0)));
    t.wcols_qchunk = std::max(0, geti("BDS_ACQ_WCOLS_QCHUNK", 0));
    t.multi_force_rccl = has("BDS_MULTI_FORCE_RCCL");
    t.trk_nblocks = std::max(0, geti("BDS_TRK_NBLOCKS", 0));
    t.trk_chunk = std::max(0, geti("BDS_TRK_CHUNK", 0));
    t.trk_persample = has("BDS_TRK_PERSAMPLE");
    t.trk_seg = geti("BDS_TRK_SEG", 0);
    t.trk_nofuse_update = has("BDS_TRK_NOFUSE_UPDATE");
#endif
    return t;
}

}  // namespace bds

extern "C" bds_ctx *bds_create(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        bds::fail(nullptr, BDS_ERR_HIP, "no HIP device visible (%s): libbds_mi355x has no CPU fallback",
                  e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        bds::fail(nullptr, BDS_ERR_ARG, "device_id %d out of range (0..%d)", device_id, n - 1);
        return nullptr;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        bds::fail(nullptr, BDS_ERR_HIP, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
        return nullptr;
    }
    bds_ctx *ctx = new bds_ctx();
    ctx->device = device_id;
    ctx->tune = bds::tuning_from_env();
    hipStream_t s;
    if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) {
        bds::fail(nullptr, BDS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    ctx->stream = (void *)s;
    hipStream_t s2;
    if ((e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)) != hipSuccess) {
        bds::fail(nullptr, BDS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        (void)hipStreamDestroy(s);
        delete ctx;
        return nullptr;
    }
    ctx->stream2 = (void *)s2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        ctx->devname = prop.name;
        ctx->devname += " ";
        ctx->devname += prop.gcnArchName;
        ctx->n_cu = prop.multiProcessorCount;
    }
    return ctx;
}

extern "C" int bds_reload_tuning(bds_ctx *ctx) {
    if (!ctx) return BDS_ERR_ARG;
    ctx->tune = bds::tuning_from_env();
    bds::acq_state_invalidate(ctx->acq);  // plan, storage mode and spectrum layout follow the knobs
    return BDS_OK;
}

extern "C" void bds_destroy(bds_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize((hipStream_t)ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize((hipStream_t)ctx->stream2);
    bds::acq_state_free(ctx->acq);
    bds::track_state_free(ctx->trk);
    if (ctx->stream2) (void)hipStreamDestroy((hipStream_t)ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy((hipStream_t)ctx->stream);
    delete ctx;
}

extern "C" const char *bds_last_error(const bds_ctx *ctx) {
    return ctx ? ctx->err.c_str() : bds::g_create_error.c_str();
}

extern "C" int bds_device_name(const bds_ctx *ctx, char *buf, int buflen) {
    if (!ctx || !buf || buflen < 1) return BDS_ERR_ARG;
    snprintf(buf, (size_t)buflen, "%s", ctx->devname.c_str());
    return BDS_OK;
}

extern "C" int bds_abi_check(int sz_settings, int sz_channel, int sz_track_out, int sz_timing) {
    return (sz_settings == (int)sizeof(bds_settings) && sz_channel == (int)sizeof(bds_channel) &&
            sz_track_out == (int)sizeof(bds_track_out) && sz_timing == (int)sizeof(bds_timing))
               ? BDS_OK
               : BDS_ERR_ARG;
}

extern "C" int bds_build_flags(void) {
    int f = 0;
#ifdef BDS_TEST_HOOKS
    f |= 1;
#endif
#ifdef BDS_DEBUG
    f |= 2;
#endif
    return f;
}

extern "C" int bds_get_timing(bds_ctx *ctx, bds_timing *t) {
    if (!ctx || !t) return BDS_ERR_ARG;
    *t = ctx->timing;
    return BDS_OK;
}

// Common/calcLoopCoef.m:41-45
extern "C" void bds_calc_loop_coef(double lbw, double zeta, double k, double *tau1, double *tau2) {
    const double wn = lbw * 8 * zeta / (4 * zeta * zeta + 1);
    if (tau1) *tau1 = k / (wn * wn);
    if (tau2) *tau2 = 2.0 * zeta / wn;
}

// Common/calcLoopCoefCarr.m:41-56  (a3 = b3 = 2, Wn = 1.2*LBW)
extern "C" void bds_calc_loop_coef_carr(const bds_settings *s, double *pf3, double *pf2, double *pf1) {
    const double wn = 1.2 * s->pllNoiseBandwidth;
    const double t = s->intTime;
    if (pf3) *pf3 = wn * wn * wn * t * t;
    if (pf2) *pf2 = 2 * wn * wn * t;
    if (pf1) *pf1 = 2 * wn;
}

// B1C/include/CalcWeighingFactor.m:43-81.  The PSDs are written with the removable
// singularities cancelled analytically:
//   sin(pi f/fc)/cos(pi f/(2fc))  = 2 sin(pi f/(2fc))
//   sin(pi f/fc)/cos(pi f/(12fc)) = 2 sum_{m=0..5} (-1)^(m+1)... = 2[sin11b - sin9b + sin7b - sin5b + sin3b - sinb], b = pi f/(12fc)
// and integrated with composite 16-point Gauss-Legendre (MATLAB integral() is adaptive
// Gauss-Kronrod; both agree to ~1e-12 relative on these smooth integrands).
static double gl16_integrate(double (*fn)(double, double, double), double fc, double tc, double a, double b, int panels) {
    static const double x[8] = {0.0950125098376374401853193, 0.2816035507792589132304605, 0.4580167776572273863424194,
                                0.6178762444026437484466718, 0.7554044083550030338951012, 0.8656312023878317438804679,
                                0.9445750230732325760779884, 0.9894009349916499325961542};
    static const double w[8] = {0.1894506104550684962853967, 0.1826034150449235888667637, 0.1691565193950025381893121,
                                0.1495959888165767320815017, 0.1246289712555338720524763, 0.0951585116824927848099251,
                                0.0622535239386478928628438, 0.0271524594117540948517806};
    double total = 0;
    const double h = (b - a) / panels;
    for (int p = 0; p < panels; ++p) {
        const double c = a + (p + 0.5) * h, r = 0.5 * h;
        double acc = 0;
        for (int i = 0; i < 8; ++i) acc += w[i] * (fn(c + r * x[i], fc, tc) + fn(c - r * x[i], fc, tc));
        total += acc * r;
    }
    return total;
}
static const double kPiD = 3.14159265358979323846;
static double g_boc11(double f, double fc, double tc) {
    if (f == 0) return 0;
    const double a = kPiD / 2 * f / fc;
    const double v = std::sin(a) * 2 * std::sin(a) * fc / f / kPiD;
    return tc * v * v;
}
static double g_boc61(double f, double fc, double tc) {
    if (f == 0) return 0;
    const double b = kPiD / 12 * f / fc;
    const double ratio = 2 * (std::sin(11 * b) - std::sin(9 * b) + std::sin(7 * b) - std::sin(5 * b) + std::sin(3 * b) - std::sin(b));
    const double v = std::sin(b) * ratio * fc / f / kPiD;
    return tc * v * v;
}
static double g_pilot(double f, double fc, double tc) { return 29.0 / 33 * g_boc11(f, fc, tc) + 4.0 / 33 * g_boc61(f, fc, tc); }
static double g_boc11_f2(double f, double fc, double tc) { return g_boc11(f, fc, tc) * f * f; }
static double g_pilot_f2(double f, double fc, double tc) { return g_pilot(f, fc, tc) * f * f; }

extern "C" double bds_calc_weighing_factor(const bds_settings *s) {
    const double fc = s->codeFreqBasis, tc = 1 / fc, br = s->FEBW;
    const int panels = 4096;
    // even integrands: 2 * integral over [0, Br/2]
    const double p11_2 = 2 * gl16_integrate(g_boc11_f2, fc, tc, 0, br / 2, panels);
    const double p11 = 2 * gl16_integrate(g_boc11, fc, tc, 0, br / 2, panels);
    const double pp_2 = 2 * gl16_integrate(g_pilot_f2, fc, tc, 0, br / 2, panels);
    const double pp = 2 * gl16_integrate(g_pilot, fc, tc, 0, br / 2, panels);
    const double rem11 = std::sqrt(p11_2 / p11), remp = std::sqrt(pp_2 / pp);
    const double t1 = 11 * p11 * rem11 * rem11, t2 = 33 * pp * remp * remp;
    return t1 / (t1 + t2);
}

// preRun.m:61-76 on the device: one wave; lane p owns PRN p + 1.  sort(peakMetric, 'descend') is stable, so the rank
// of PRN p is the number of PRNs with a larger metric plus the number of EARLIER PRNs with an equal one.
__global__ void k_pre_run(int max_prn, int nch, int b1c, double codeFreqBasis, double IF, double carrFreqBasis,
                          const double *__restrict__ carrFreq, const double *__restrict__ codePhase,
                          const double *__restrict__ peakMetric, bds_channel *__restrict__ channel) {
    __shared__ double pm[64];
    __shared__ int ndet_s;
    const int p = (int)threadIdx.x;
    if (p == 0) ndet_s = 0;
    pm[p] = p < max_prn ? peakMetric[p] : 0.0;
    __syncthreads();
    if (p < nch) {
        bds_channel c;
        c.PRN = 0;
        c.status = '-';
        c.acquiredFreq = c.codePhase = c.codeFreq = 0;
        channel[p] = c;
    }
    if (p < max_prn && carrFreq[p] != 0) atomicAdd(&ndet_s, 1);
    __syncthreads();
    if (p >= max_prn) return;
    int rank = 0;
    for (int j = 0; j < max_prn; ++j) rank += (pm[j] > pm[p]) || (pm[j] == pm[p] && j < p);
    const int ndet = ndet_s;
    if (rank < (nch < ndet ? nch : ndet)) {
        bds_channel c;
        c.PRN = p + 1;
        c.acquiredFreq = carrFreq[p];
        c.codePhase = codePhase[p];
        // B1C/include/preRun.m:71-73 (Doppler aiding) / B2a/include/preRun.m:70; the division before the product, as
        // the reference writes it (this file is compiled with -ffp-contract=off)
        c.codeFreq = b1c ? codeFreqBasis - (c.acquiredFreq - IF) / carrFreqBasis * codeFreqBasis : codeFreqBasis;
        c.status = 'T';
        channel[rank] = c;
    }
}

extern "C" int bds_pre_run_device(bds_ctx *ctx, const bds_settings *s, int max_prn, const double *carrFreq,
                                  const double *codePhase, const double *peakMetric, bds_channel *channel) {
    if (!ctx || !s || !carrFreq || !codePhase || !peakMetric || !channel || max_prn < 1 || max_prn > BDS_MAX_PRN) return BDS_ERR_ARG;
    const int nch = s->numberOfChannels;
    if (nch < 1) return bds::fail(ctx, BDS_ERR_ARG, "numberOfChannels = %d < 1", nch);
    // (the kernel is one wave: it fills up to 64 channels -- one more than there are PRNs; a larger table is filled by the
    //  host loop, which produces the same bits)
    if (nch > 64) return bds_pre_run(s, max_prn, carrFreq, codePhase, peakMetric, channel);
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)ctx->stream;
    double *d_in = nullptr;
    bds_channel *d_ch = nullptr;
    struct Scope {
        void **a, **b;
        ~Scope() {
            if (*a) (void)hipFree(*a);
            if (*b) (void)hipFree(*b);
        }
    } scope{(void **)&d_in, (void **)&d_ch};
    BDS_HIP(ctx, hipMalloc((void **)&d_in, sizeof(double) * 3 * max_prn));
    BDS_HIP(ctx, hipMalloc((void **)&d_ch, sizeof(bds_channel) * nch));
    BDS_HIP(ctx, hipMemcpyAsync(d_in, carrFreq, sizeof(double) * max_prn, hipMemcpyHostToDevice, st));
    BDS_HIP(ctx, hipMemcpyAsync(d_in + max_prn, codePhase, sizeof(double) * max_prn, hipMemcpyHostToDevice, st));
    BDS_HIP(ctx, hipMemcpyAsync(d_in + 2 * max_prn, peakMetric, sizeof(double) * max_prn, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pre_run, dim3(1), dim3(64), 0, st, max_prn, nch, s->signal == BDS_SIGNAL_B1C ? 1 : 0, s->codeFreqBasis, s->IF,
                       s->carrFreqBasis, (const double *)d_in, (const double *)(d_in + max_prn), (const double *)(d_in + 2 * max_prn), d_ch);
    BDS_HIP(ctx, hipGetLastError());
    BDS_HIP(ctx, hipMemcpyAsync(channel, d_ch, sizeof(bds_channel) * nch, hipMemcpyDeviceToHost, st));
    BDS_HIP(ctx, hipStreamSynchronize(st));
    return BDS_OK;
}

extern "C" int bds_acquire_track(bds_ctx *ctx, const bds_settings *s, const int8_t *samples, size_t n_samples, int is_complex,
                                 int max_prn, double *carrFreq, double *codePhase, double *peakMetric, int32_t *detected,
                                 const char *path, bds_channel *channel, bds_track_out *out) {
    if (!ctx || !s || !path || !channel || !out) return BDS_ERR_ARG;
    int rc = bds_acquire(ctx, s, samples, n_samples, is_complex, max_prn, carrFreq, codePhase, peakMetric, detected);
    if (rc) return rc;
    if ((rc = bds_pre_run_device(ctx, s, max_prn, carrFreq, codePhase, peakMetric, channel))) return rc;
    return bds_track(ctx, s, path, s->numberOfChannels, channel, out);
}

// preRun.m:61-76
extern "C" int bds_pre_run(const bds_settings *s, int max_prn, const double *carrFreq, const double *codePhase,
                           const double *peakMetric, bds_channel *channel) {
    if (!s || !carrFreq || !codePhase || !peakMetric || !channel || max_prn < 1) return BDS_ERR_ARG;
    const int nch = s->numberOfChannels;
    for (int i = 0; i < nch; ++i) {
        channel[i].PRN = 0;
        channel[i].status = '-';
        channel[i].acquiredFreq = channel[i].codePhase = channel[i].codeFreq = 0;
    }
    std::vector<int> idx((size_t)max_prn);
    std::iota(idx.begin(), idx.end(), 0);
    // sort(peakMetric, 'descend') -- stable, ties keep ascending PRN order
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return peakMetric[a] > peakMetric[b]; });
    int ndet = 0;
    for (int i = 0; i < max_prn; ++i) ndet += carrFreq[i] != 0;
    for (int ii = 0; ii < std::min(nch, ndet); ++ii) {
        const int p = idx[(size_t)ii];
        channel[ii].PRN = p + 1;
        channel[ii].acquiredFreq = carrFreq[p];
        channel[ii].codePhase = codePhase[p];
        if (s->signal == BDS_SIGNAL_B1C)  // B1C/include/preRun.m:71-73
            channel[ii].codeFreq = s->codeFreqBasis - (channel[ii].acquiredFreq - s->IF) / s->carrFreqBasis * s->codeFreqBasis;
        else  // B2a/include/preRun.m:70
            channel[ii].codeFreq = s->codeFreqBasis;
        channel[ii].status = 'T';
    }
    return BDS_OK;
}

#ifdef BDS_DEBUG
// debug build only (csrc/bds_debug.h): failed device-side bounds checks since the last call, all translation units
extern "C" unsigned int bds_debug_failures_acq();
extern "C" unsigned int bds_debug_failures_track();
extern "C" __attribute__((visibility("default"))) unsigned int bds_debug_failures(void) {
    return bds_debug_failures_acq() + bds_debug_failures_track();
}
#endif
