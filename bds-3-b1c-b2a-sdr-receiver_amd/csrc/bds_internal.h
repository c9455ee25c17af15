// Internal declarations shared by the translation units of libbds_mi355x.so.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "bds_mi355x.h"

namespace bds {

// host code generation (bds_codes.cpp)
int gen_primary(int signal, bool pilot, int prn, int8_t *out /*10230*/);
int gen_secondary(int prn, int8_t *out /*1800*/);

// MATLAB round(): half away from zero
inline double m_round(double x) { return x >= 0 ? (double)(long long)(x + 0.5) : -(double)(long long)(-x + 0.5); }

// samplesPerCode = round(fs / (codeFreqBasis / codeLength))   (B2a/acquisition.m:130-131)
inline long samples_per_code(const bds_settings &s) {
    return (long)m_round(s.samplingFreq / (s.codeFreqBasis / s.codeLength));
}

// Tuning / test knobs, read from the environment ONCE at bds_create and kept per context (nothing on the
// product path calls getenv afterwards).  Defaults are the measured best; tools/README.md lists them.
struct Tuning {
    int force_l1 = 0, force_l2 = 0;  // BDS_ACQ_FORCE_L1L2=AxB: two-pass factorisation
    int logt = -1;                   // BDS_ACQ_LOGT: log2 of the column tile width
    bool generic = false;            // BDS_ACQ_GENERIC: run-time-radix search kernels
    bool generic_fwd = false;        // BDS_ACQ_GENERIC_FWD: run-time-radix forward transforms
    int group = 0;                   // BDS_ACQ_GROUP: (PRN, bin) cells per launch pair (0 = whole Doppler row)
    int fp16_storage = -1;           // BDS_ACQ_FP16: spectra / inter-pass buffer as fp16 complex (-1 = default on)
    int gchunk = 34;                 // BDS_ACQ_GCHUNK: cells one row-pass workgroup walks through
    bool multi_any = false, nomulti = false;  // BDS_ACQ_MULTI_ANY / BDS_ACQ_NOMULTI: multi-PRN launch pairs
    int pbcells = 0;                 // BDS_ACQ_PBCELLS
    double pbcap_gb = 0;             // BDS_ACQ_PBCAP_GB (hooks): budget of the inter-pass buffer of a multi-PRN launch pair, overrides pair_gb
    bool pair_gb_env = false;        // BDS_ACQ_PAIR_GB was given
    double pair_gb = 40;             // BDS_ACQ_PAIR_GB (release knob): budget of the inter-pass buffer in GiB -- a launch pair carries as many PRNs' Doppler
                                     // rows as fit (default 40: 8 PRNs at cfg3); 0 = minimal (one PRN's row per pair on big grids; small grids batch up
                                     // to 8 GiB), < 0 ("auto") = 60 % of the free device memory
    int rows_grid = 0;                  // BDS_ACQ_ROWS_GRID: workgroups of the (then persistent) row pass; 0 = one per item
    int list_gc = 0;                    // BDS_ACQ_LIST_GC: cells one row workgroup walks in a multi-PRN launch pair (a divisor of D; 0 = chosen by the launch's size, -1 = all D bins of its PRN)
    bool no_bwreuse = false;            // BDS_ACQ_NO_BWREUSE: the B2a second-peak pass runs its own row pass (A/B, tests)
    bool overlap = false;               // BDS_ACQ_OVERLAP: column pass of group k on a second stream beside the row pass of group k+1
    double kdelta = 0;                  // BDS_ACQ_KDELTA: test hook, sieve tolerance override (0 = per-mode default)
    bool no_selfcheck = false;          // BDS_ACQ_NO_SELFCHECK: timing experiments with invalid results (no re-run)
    bool test_force_fallback = false;   // BDS_ACQ_TEST_FORCE_FALLBACK: test hook, take the fp16 -> fp32 storage re-run
    int wcols = -1;                  // BDS_ACQ_WCOLS: wave-private column pass (bds_acq_wcols.h); -1 = default on, 0 = the round-2 tile kernel
    int clockprobe = 0;              // BDS_ACQ_CLOCKPROBE: sampled workgroups of the wave-private search kernels time themselves (bds_timing::shader_clock_GHz)
    int wrows = -1;                  // BDS_ACQ_WROWS: wave-private 4096-point row pass (bds_acq_wrows.h); -1 = default on, 0 = k_rows_inv_f
    int pfa = 1;                     // BDS_ACQ_PFA (hooks): the N-point search pair of bds_acq_pfa.h where it applies (B1C, N = 53 x 12 x 3125, fp16 storage,
                                     // whole bins per acqStep); 0 = the L-point pair of rounds 3-5 everywhere
    int pfa_qchunk = 0;              // BDS_ACQ_PFA_QCHUNK (hooks): blocks of 16 lags of a cell that follow each other in the N-point column pass's work list
    int pfa_cgrid = 0;               // BDS_ACQ_PFA_CGRID (hooks): workgroups of the N-point column pass
    int small_plan = 1;              // BDS_ACQ_SMALL: small two-component searches on the 80 x 4096 plan (bds_acq_scols.h); 0 = 256 x 1280 as in rounds 1-3
    bool host_refine = false;        // BDS_ACQ_HOSTREFINE: refinement through the host (lists downloaded, jobs built there: rounds 1-4) instead of the device chain
    int neigh = 0;                   // BDS_ACQ_NEIGH: also refine the +-n bin / lag neighbours of every candidate in f64 (rounds 1-3: 1)
    int pk = 1;                      // BDS_ACQ_PK: packed-fp32 butterflies in the wave-private search kernels (bds_fft_pk.h); 0 = one fp32 instruction per real operation
    int ilv = 1;                     // BDS_ACQ_ILV: the wave-private pair of the 768 x 4096 plan keeps both components of an element side by side in the inter-pass buffer (0 = separate planes)
    int wcols_qchunk = 0;            // BDS_ACQ_WCOLS_QCHUNK: adjacent 128-byte lines of a cell its work list keeps together (DRAM page locality; round 3, one PRN per launch:
                                     // 1: 1.84, 2: 1.70, 4: 1.70, 8: 1.75, 16: 1.83 ms per cfg3 launch); 0 = 4 for a launch of one PRN's cells, 2 for a multi-PRN launch (round 5, 32 PRNs
                                     // per pair, per call: 4: 189.0-190.1, 2: 187.9-188.3, 3: 187.9-188.5, 1: 188.5-188.6, 8: 191.4 ms; one PRN per pair: 4 and 2 the same, 1 +1 %)
    bool verbose = false;            // BDS_VERBOSE
    bool multi_force_rccl = false;      // BDS_MULTI_FORCE_RCCL: a single-device bds_multi still goes through RCCL (test hook)
    int trk_nblocks = 0;                // BDS_TRK_NBLOCKS: test hook, correlate workgroups per channel (0 = sized from the code rate)
    int trk_chunk = 0;               // BDS_TRK_CHUNK: samples per correlate workgroup (0 = per-mode default)
    bool trk_nofuse_update = false;  // BDS_TRK_NOFUSE_UPDATE: loop update as its own launch per epoch instead of at the head of the next correlate launch
    bool trk_persample = false;      // BDS_TRK_PERSAMPLE: per-sample tracking correlator instead of the run-based one
    int trk_prec = 4;                // BDS_TRK_PREC: carrier / prefix-sum numerics of the run-based correlator (bds_track.hip, TrkParams::prec; 4 = strict, the default: a sin / cos of the reference's trigarg per sample; 5 = the same argument by angle addition)
    int trk_seg = 0;                 // BDS_TRK_SEG: samples per lane and pass of the run-based correlator (8 / 16; 0 = per-signal default)
};
Tuning tuning_from_env();

struct AcqState;    // bds_acq.hip
struct TrackState;  // bds_track.hip
void acq_state_free(AcqState *);
void acq_state_invalidate(AcqState *);  // forget the configuration (plan, storage mode, cached spectra): re-derived by the next prepare
void track_state_free(TrackState *);

}  // namespace bds

struct bds_ctx {
    int device = 0;
    void *stream = nullptr;   // hipStream_t: everything is ordered on this stream ...
    void *stream2 = nullptr;  // ... except the search's column pass, which overlaps the next row pass
    std::string err;
    std::string devname;
    bds::AcqState *acq = nullptr;
    bds::TrackState *trk = nullptr;
    bds_timing timing{};
    bds::Tuning tune;
    std::set<const void *> lds_attr_done;  // kernels whose dynamic-LDS limit has been raised on this device
    int n_cu = 0;                          // compute units of the device
};

namespace bds {
int fail(bds_ctx *ctx, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

// Stage markers for `rocprofv3 --marker-trace` (the reference's own stage timing is tic / toc around
// acquisition(), B1C/postProcessing.m:104,112): roctxRangePush / Pop around the forward pass, the search, the refinement and the
// tracking epoch loop.  The ROCTx library is resolved with dlopen at the first range (bds_api.hip) -- no link dependency; without it,
// or outside a profiler, a range costs one predictable branch.
void roctx_push(const char *name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char *name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};
}

#define BDS_HIP(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return bds::fail((ctx), BDS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                             hipGetErrorString(_e), __FILE__, __LINE__);                       \
    } while (0)
