/*
 * bds_mi355x.h -- C ABI of libbds_mi355x.so: MI355X (gfx950) acquisition and
 * tracking correlators for BDS-3 B1C / B2a.
 *
 * The reference (lyf8118/BDS-3-B1C-B2a-SDR-receiver) is pure MATLAB and has no
 * FFI layer of its own; the drop-in boundary is the MATLAB call surface used by
 * postProcessing.m.  Every entry point below names the reference interface it
 * replaces (paths relative to /root/reference/BDS3_B1C_B2a).  The MEX gateway
 * (mex/bds_mex.c) and the ctypes host layer (bds_amd/native.py) bind exactly
 * these symbols.  Plain pointers and sizes only; all buffers are caller-owned
 * host memory unless stated.
 *
 * Conventions: every function returning int returns 0 on success, <0 on error
 * (BDS_ERR_*); bds_last_error() gives the message.  Indices that the reference
 * reports 1-based (codePhase) stay 1-based; absoluteSample stays 0-based.
 */
#ifndef BDS_MI355X_H
#define BDS_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDS_API __attribute__((visibility("default")))

#define BDS_MAX_PRN 63

enum { BDS_SIGNAL_B1C = 1, BDS_SIGNAL_B2A = 2 };
/* tracking variant: B2a/tracking.m | B1C/NB_tracking.m | B1C/WB_tracking.m */
enum { BDS_TRACK_B2A = 0, BDS_TRACK_NB = 1, BDS_TRACK_WB = 2 };
/* bds_gen_code kinds */
enum {
    BDS_CODE_DATA_PRIMARY = 0,  /* 10230 chips (+-1)                                        */
    BDS_CODE_PILOT_PRIMARY = 1, /* 10230 chips                                               */
    BDS_CODE_DATA_BOC11 = 2,    /* B1C only, 20460 half-chips   generateDataBOC11.m:85-91    */
    BDS_CODE_PILOT_BOC11 = 3,   /* B1C only, 20460              generatePilotBOC11.m:88-94   */
    BDS_CODE_PILOT_BOC61 = 4,   /* B1C only, 122760             generatePilotBOC61.m:89-96   */
    BDS_CODE_PILOT_SECONDARY = 5 /* B1C only, 1800 chips         generate2ndCode.m:59-84      */
};

enum {
    BDS_OK = 0,
    BDS_ERR_ARG = -1,         /* bad argument / settings field                    */
    BDS_ERR_HIP = -2,         /* HIP runtime error                                */
    BDS_ERR_UNSUPPORTED = -3, /* valid in the reference, not built yet            */
    BDS_ERR_IO = -4,          /* file open/read failure (fopen error, B2a/postProcessing.m:152-154) */
    BDS_ERR_NOMEM = -5
};

/*
 * Flat mirror of the `settings` struct fields the acquisition/tracking path
 * reads (B1C/initSettings.m:48-151, B2a/initSettings.m:44-130).
 * Field names are the MATLAB names.
 */
typedef struct bds_settings {
    int32_t signal; /* BDS_SIGNAL_*: which receiver directory the struct came from */
    int32_t fileType;              /* 1 = real int8, 2 = interleaved I/Q int8        */
    double samplingFreq;           /* [Hz] */
    double IF;                     /* [Hz] */
    double codeFreqBasis;          /* [Hz] */
    double carrFreqBasis;          /* [Hz] (B1C preRun Doppler aiding)               */
    int32_t codeLength;            /* [chips] 10230                                  */
    int32_t numberOfChannels;
    int64_t skipNumberOfBytes;
    double msToProcess;            /* [ms] */
    /* acquisition */
    double acqSearchBand;          /* [Hz] half width                                */
    double acqStep;                /* [Hz] */
    double acqThreshold;
    double acqCohT;                /* [ms] B1C only                                  */
    int32_t pilotACQflag;          /* B1C only                                       */
    int32_t fineNoncoh;            /* B2a only: code periods in the fine search      */
    double resamplingThreshold;
    int32_t resamplingflag;        /* 1: condition + resample when fs > threshold   */
    int32_t n_acq;                 /* length of acqSatelliteList                     */
    int32_t acqSatelliteList[BDS_MAX_PRN];
    /* tracking */
    int32_t pilotTRKflag;          /* B2a 0/1; B1C 0 / 1 (NB) / 2 (WB)               */
    double intTime;                /* [s] */
    double dllCorrelatorSpacing;   /* [chips] */
    double dllDampingRatio;
    double dllNoiseBandwidth;      /* [Hz] */
    double pllNoiseBandwidth;      /* [Hz] */
    int32_t CNoInterval;
    int32_t dataType;              /* settings.dataType: 0 = 'schar' (int8 samples, the only format the tracking
                                      scripts' fread(..., dataType) is ever called with here); anything else is
                                      rejected with BDS_ERR_UNSUPPORTED (B2a/tracking.m:237-238)            */
    double FEBW;                   /* [Hz] B1C WB only (CalcWeighingFactor.m:46)     */
} bds_settings;

/* channel(1 x nCh) of preRun.m:46-56 */
typedef struct bds_channel {
    int32_t PRN;          /* 0 = unused */
    int32_t status;       /* '-' or 'T' */
    double acquiredFreq;  /* [Hz] */
    double codePhase;     /* 1-based sample */
    double codeFreq;      /* [Hz] */
} bds_channel;

/*
 * trackResults(1 x nCh) of B2a/tracking.m:48-96 / B1C/WB_tracking.m:53-112 as
 * structure-of-arrays: every pointer is caller-allocated double[n_ch * n_epochs]
 * (channel-major) except the C/N0 arrays, double[n_ch * n_cno], and
 * `completed`/`status`.  Pointers of fields the selected variant does not create
 *  may be NULL.  The library initialises the arrays the way
 * the reference template does (zeros / inf) before tracking.
 */
typedef struct bds_track_out {
    int32_t n_ch, n_epochs, n_cno, reserved0;
    double *absoluteSample, *codeFreq, *carrFreq;
    double *I_P, *I_E, *I_L, *Q_E, *Q_P, *Q_L;
    double *Pilot_I_P, *Pilot_Q_P;                         /* pilot tracking on     */
    double *Pilot_I_E, *Pilot_I_L, *Pilot_Q_E, *Pilot_Q_L; /* WB only               */
    double *dllDiscr, *dllDiscrFilt, *pllDiscr, *pllDiscrFilt;
    double *remCodePhase, *remCarrPhase;
    double *DataCNo, *DataPLD, *PilotCNo, *PilotPLD, *SigCNo; /* SigCNo = B1C_CNo | B2a_CNo */
    int32_t *completed; /* [n_ch] epochs finished per channel                        */
    int32_t *status;    /* [n_ch] '-' or 'T' (B2a/tracking.m:48,441)                 */
} bds_track_out;

/* per-stage device timing of the last bds_acq_run / bds_track call (hipEvent, ms) */
typedef struct bds_timing {
    double total_ms;        /* whole call, stream time                              */
    double forward_ms;      /* wipe-off + forward transforms, all bins              */
    double search_ms;       /* all (PRN, bin) cells: multiply + inverse + |.| max   */
    double refine_ms;       /* f64 re-evaluation of candidate cells + fine search   */
    double cell_pair_ms;    /* average duration of one (rows+cols) launch pair, sampled */
    double cells_per_pair;  /* (PRN, bin) cells one launch pair processes           */
    int64_t n_pairs;        /* launch pairs in the call                             */
    int64_t fft_len;        /* padded transform length L                            */
    int64_t n_circ;         /* N: the reference's circular correlation length       */
    int32_t n_bins, n_prn, n_comp;
    int32_t half_storage;   /* 0: fp32 search on fp32 storage; 1: fp32 arithmetic, spectra + inter-pass buffer
                               held as fp16 complex (default)                                           */
    double rows_ms;         /* average duration of the row-pass kernel of a sampled launch pair        */
    double cols_ms;         /* ... and of its column-pass kernel                                       */
    int64_t n_extra;        /* entries of the sieve's overflow list in the last search                 */
    double shader_clock_GHz; /* engine clock the search kernels ran at (sampled workgroups time themselves with the shader
                                clock against the reference clock); 0 unless BDS_ACQ_CLOCKPROBE=1              */
    int32_t plan_l1, plan_l2; /* two-pass factorisation of fft_len: column length x row length                          */
    int32_t rows_kernel;    /* row pass of the search: 0 run-time plan (k_rows_inv), 1 k_rows_inv_f, 2 k_rows_wave_f, 3 k_pfa_rows (N-point pair) */
    int32_t cols_kernel;    /* column pass: 0 run-time plan (k_cols_inv_max), 1 tile kernel k_cols_inv_max_f, 2 k_cols_wave_f, 3 k_cols_small_f, 4 k_pfa_cols */
    int32_t kernel_flags;   /* bit 0: components interleaved in the inter-pass buffer; bit 1: packed-fp32 butterflies     */
    int32_t refine_path;     /* 1: candidates -> f64 sums -> peak / second peak / fine search as one device chain with a single download (csrc/bds_acq_refine.h); 0: through the host */
} bds_timing;

typedef struct bds_ctx bds_ctx;

/* ---- context ---------------------------------------------------------------- */
/* One context per GPU (one process per GPU under torch.distributed / RCCL). */
BDS_API bds_ctx *bds_create(int device_id);
BDS_API void bds_destroy(bds_ctx *ctx);
/* Environment knobs, read once at bds_create into the context.  The release library reads exactly these:
 *   BDS_ACQ_FP16=0       fp32 storage of the spectra and the inter-pass buffer as well (default: fp16 storage; every reported
 *                        value is decided in f64 either way)
 *   BDS_TRK_PREC=0..5    numerics of the tracking correlator: 4 (default) a sin / cos of the reference's own carrier argument
 *                        trigarg(k) per sample in f64 -- correlator sums 1e-13 of |P| from the float64 oracle, the loop state
 *                        bit-identical for hundreds of epochs, SURVEY.md 8d until a channel's first ceil() flip (cfg4 at full
 *                        rate: 11 of 12 channels over all 3 600 epochs); 5: the same argument by angle addition (one sin / cos
 *                        per 16 samples), 12 % faster, 4e-10 of |P| (6 of 12 channels); 0: fp32 carrier recurrence, ~1.5x
 *                        faster in wide-band mode, first flip after a few hundred epochs
 *   BDS_VERBOSE          progress / fallback messages on stderr
 *   BDS_ACQ_CLOCKPROBE=1 sampled workgroups time themselves with the shader clock (bds_timing.shader_clock_GHz)
 *   BDS_ACQ_PAIR_GB=n|auto budget of the search's inter-pass buffer in GiB: several PRNs' Doppler rows per launch pair (default 40;
 *                        "auto": 60 % of the free device memory, the serving mode; 0: minimal).  See bds_acq_set_pair_budget_gb.
 * (plus BDS_MEX_DEVICES in the MEX gateway and BDS_LIB_PATH in the ctypes host).  Kernel-selection, launch-shape and sieve
 * switches, the RCCL path override and the aliased-device hook exist only in libbds_mi355x_hooks.so (-DBDS_TEST_HOOKS,
 * built beside the release library by build.sh; tools/README.md lists them), which is what tests/ load.
 * bds_reload_tuning re-reads the knobs into an existing context and makes the next bds_acq_prepare re-derive the
 * acquisition configuration (plan, storage mode, cached code spectra).  Never needed by a host application. */
BDS_API int bds_reload_tuning(bds_ctx *ctx);
BDS_API const char *bds_last_error(const bds_ctx *ctx); /* ctx may be NULL: creation errors */
BDS_API int bds_device_name(const bds_ctx *ctx, char *buf, int buflen);
/* Binding self-check: returns 0 when the caller's struct sizes equal the library's. */
BDS_API int bds_abi_check(int sizeof_settings, int sizeof_channel, int sizeof_track_out, int sizeof_timing);
/* Which build this is: bit 0 = test hooks compiled in (libbds_mi355x_hooks.so), bit 1 = debug build (device-side bounds checks). */
BDS_API int bds_build_flags(void);

/* ---- ranging codes (host side, no GPU needed) -------------------------------- */
/* Replaces generateB2aDataCode.m / generateB2aPilotCode.m / generateDataBOC11.m /
 * generatePilotBOC11.m / generatePilotBOC61.m.  Writes +-1 into out[0..n); returns
 * the code length, or <0.  n must be >= the code length. */
BDS_API int bds_gen_code(int signal, int kind, int prn, int8_t *out, int n);

/* ---- acquisition -------------------------------------------------------------
 * acqResults = acquisition(longSignal, settings)
 *   B2a/acquisition.m:1, B1C/acquisition.m:1, B1C/GPU_acquisition.m:1
 * samples: int8 IF samples as fread(...,'schar') delivers them
 *   (B2a/postProcessing.m:89-90, B1C/postProcessing.m:94); n_samples real samples
 *   (or I/Q pairs when is_complex, fileType 2).
 * carrFreq/codePhase/peakMetric: double[max_prn], max_prn >= max(acqSatelliteList);
 *   zero where not searched / not detected (B2a/acquisition.m:161-165).
 * detected (optional, may be NULL): int32[max_prn], 1 where the PRN passed the
 *   threshold -- lets the caller print the reference's "(19 20 . )" line
 *   (B2a/acquisition.m:167,259,360,366).
 */
BDS_API int bds_acquire(bds_ctx *ctx, const bds_settings *s, const int8_t *samples,
                        size_t n_samples, int is_complex, int max_prn, double *carrFreq,
                        double *codePhase, double *peakMetric, int32_t *detected);

/* The same call split in three so that the timed region of a benchmark starts with
 * the IF block resident in HBM and the code spectra cached:
 *   bds_acq_load    H2D copy of the IF block
 *   bds_acq_prepare code spectra for settings.acqSatelliteList (cached in ctx)
 *   bds_acq_run     forward transforms, PRN x Doppler search, refinement, fine search
 * prn_list/n_prn select the PRN shard this rank searches (NULL/0 = acqSatelliteList);
 * outputs are zero for PRNs outside the shard so that an all-reduce(SUM) across
 * ranks reassembles acqResults bit-exactly (x + 0). */
BDS_API int bds_acq_load(bds_ctx *ctx, const bds_settings *s, const int8_t *samples,
                         size_t n_samples, int is_complex);
BDS_API int bds_acq_prepare(bds_ctx *ctx, const bds_settings *s);
BDS_API int bds_acq_run(bds_ctx *ctx, const bds_settings *s, const int32_t *prn_list, int n_prn,
                        int max_prn, double *carrFreq, double *codePhase, double *peakMetric,
                        int32_t *detected);
/* Budget of the search's inter-pass buffer: a launch pair (row pass + column pass) carries as many PRNs' Doppler rows as fit
 * `gib` GiB; a call is then a few long launches.  cfg3 (63 PRNs x 201 bins) on one box, round 6 (the N-point pair of
 * csrc/bds_acq_pfa.h: 3.3 GB per PRN; profiles/r06_pfa53_knobs_tiled.txt):
 *     0        one PRN per pair,  3.3 GB    144.6 ms per call     (the minimal footprint)
 *    20        5 - 6 PRNs per pair          132.8
 *    40        12 - 13 PRNs per pair        132.1 - 133.0         (the DEFAULT)
 *    80        21 PRNs per pair             132.8
 *   < 0       60 % of the device memory that is free: 32 + 31 PRNs, 100 GiB    132.2   (the SERVING mode; bench.py key `serving`)
 * With the L-point pair of rounds 3-5 (5 GB per PRN; every configuration the N-point pair does not cover) more PRNs per pair also
 * shared the 2.5 GB of signal-spectrum rows in L2: 196.7 / 191.6 - 192.1 (8 PRNs) / 189.3 - 189.8 / 186.8 - 187.1 ms.
 * The price is the footprint, and time when it changes hands: a fresh allocation is free (a first call costs the same in every
 * mode), but the driver clears freed device memory at ~33 GB/s and whoever allocates next waits -- up to ~4.8 s after a 150-GiB
 * context is destroyed.  Results are the same bits in every mode.  Same switch as the environment knob BDS_ACQ_PAIR_GB (number
 * of GiB, or "auto"); takes effect at the next bds_acq_run -- in both directions: a context whose buffer is more than a quarter
 * (+256 MiB) larger than the new budget needs frees it and allocates the smaller one at that run (round 6; until then only
 * bds_destroy returned the memory). */
BDS_API int bds_acq_set_pair_budget_gb(bds_ctx *ctx, double gib);

/* Diagnostics of the last bds_acq_run: per searched PRN (in search order) and Doppler
 * bin, the maximum of results(bin,:) (fp32 search value) and its 1-based lag.
 * row_max/row_arg: [n_prn * n_bins].  Returns n_prn*n_bins or <0. */
BDS_API int bds_acq_grid(bds_ctx *ctx, float *row_max, int32_t *row_arg, int cap);
/* Diagnostics of the sieve: the (Doppler bin, code phase) cells of `prn` the last bds_acq_run re-evaluated in
 * f64 (1-based, like the reference's indices into results(bin, codePhase)).  Returns their number (may exceed
 * cap).  Tests use it to check that every cell within the sieve tolerance of the maximum was refined. */
BDS_API int bds_acq_candidates(bds_ctx *ctx, int prn, int32_t *bin, int64_t *lag, int cap);
/* Peak / second-peak (B2a) or peak / sigPower (B1C) of the last run, per PRN slot:
 * peak[max_prn], denom[max_prn], fbin[max_prn] (1-based frequency bin). */
BDS_API int bds_acq_peaks(bds_ctx *ctx, int max_prn, double *peak, double *denom, int32_t *fbin);
/* Check entry: the f64 coherent sums the decisions rest on, for caller-chosen cells of the block loaded by the last
 * bds_acq_load / bds_acq_run (csrc/bds_acq_corr.h), as interleaved (re, im) pairs.  Returns the number of pairs or < 0.
 *   mode 0  coarse cell (B2a/acquisition.m:194-209, B1C/acquisition.m:198-219): for each of freqs[0 .. nf) and each component,
 *           sum_n x[(phase-1+n) mod N] code_c(n) exp(+j 2 pi f t/fs), t = the wrapped sample index:  out[(f*ncomp + c)*2]
 *   mode 1  fine-search block starting at code phase `phase` (B2a :287-316: fineNoncoh segments, both components, long code;
 *           B1C :253-287: one code period, DC removed), every frequency of freqs[] in the multi-frequency pass:
 *           out[((seg*ncomp + c)*nf + f)*2]
 *   mode 2  the same sums, one frequency per pass (what mode 1 must agree with to rounding) */
BDS_API int bds_acq_coherent_sums(bds_ctx *ctx, const bds_settings *s, int prn, int64_t phase, const double *freqs, int nf, int mode,
                                  double *out, int cap); /* cap: (re, im) pairs `out` holds; BDS_ERR_ARG, nothing written, if the call needs more */
BDS_API int bds_get_timing(bds_ctx *ctx, bds_timing *t);

/* ---- multi-device acquisition -------------------------------------------
 * One host process drives N GPUs: what postProcessing.m's single acquisition() call (B1C/postProcessing.m:105-111,
 * B2a/postProcessing.m:100) becomes on a multi-GPU node.  The (signal, PRN) jobs are spread over the devices by cost
 * (bds_shard_jobs: longest-processing-time rule; bds_acq_job_cost: transform points x Doppler bins x components of
 * one PRN), every device searches its shard with the whole IF block, and one RCCL all-reduce(SUM) of
 * 3 x max_prn f64 per signal over xGMI leaves the complete acqResults everywhere (x + 0: bit-identical to one
 * device).  Several signals in one call = BASELINE.json configs[4] (B1C + B2a jointly).
 * Tracking needs no exchange: use bds_multi_ctx(m, i) with bds_track per device (replicas / channel shards). */
typedef struct bds_multi bds_multi;
typedef struct bds_acq_job {
    const bds_settings *settings; /* one receiver's settings (signal, acqSatelliteList, ...)          */
    const int8_t *samples;        /* its IF block (host), as for bds_acquire                           */
    size_t n_samples;
    int32_t is_complex;
    int32_t max_prn;              /* >= max(acqSatelliteList), <= 63                                   */
    double *carrFreq, *codePhase, *peakMetric; /* out: double[max_prn] each                            */
    int32_t *detected;            /* out, optional: int32[max_prn]                                     */
} bds_acq_job;
/* n_devices <= 0: every visible device; device_ids NULL: 0 .. n_devices-1 */
BDS_API bds_multi *bds_multi_create(int n_devices, const int *device_ids);
BDS_API void bds_multi_destroy(bds_multi *m);
BDS_API const char *bds_multi_last_error(const bds_multi *m); /* m may be NULL: creation errors */
BDS_API int bds_multi_size(const bds_multi *m);
BDS_API bds_ctx *bds_multi_ctx(bds_multi *m, int i);
/* communicator size RCCL reported at the last all-reduce (0: none has run) */
BDS_API int bds_multi_rccl_ranks(const bds_multi *m);
BDS_API int bds_acquire_multi(bds_multi *m, int n_signals, const bds_acq_job *signals);
/* rank_of_job[j] in 0..world-1 for jobs of relative cost[j] (no GPU needed) */
BDS_API int bds_shard_jobs(int n_jobs, const double *cost, int world, int32_t *rank_of_job);
BDS_API double bds_acq_job_cost(const bds_settings *s);

/* ---- tracking -----------------------------------------------------------------
 * [trackResults, channel] = tracking(fid, channel, settings)
 *   B2a/tracking.m:1, B1C/NB_tracking.m:1, B1C/WB_tracking.m:1
 * The reference seeks absolutely from 'bof' (B2a/tracking.m:151-153), so the MATLAB
 * wrapper passes the file *path* (fopen(fid)) instead of the handle.
 * Returns 0 also when the file ends early: like B2a/tracking.m:250-254 the results
 * gathered so far are returned, `completed[ch]` says how far each channel got and
 * status stays '-' for the channel that hit EOF and all later ones.
 */
BDS_API int bds_track(bds_ctx *ctx, const bds_settings *s, const char *path, int n_ch,
                      const bds_channel *channel, bds_track_out *out);
/* Same, on an IF record already in host memory (n_bytes raw file bytes). */
BDS_API int bds_track_mem(bds_ctx *ctx, const bds_settings *s, const int8_t *file_bytes,
                          size_t n_bytes, int n_ch, const bds_channel *channel,
                          bds_track_out *out);
/* Bytes of the IF record the last bds_track / bds_track_mem call copied to HBM: only the window the channels can
 * touch (earliest start sample .. latest start + msToProcess at a code rate 2 % low) is loaded, so a recording far
 * longer than msToProcess (or than HBM) tracks fine; end-of-file is still judged against the real file size. */
BDS_API long long bds_track_loaded_bytes(bds_ctx *ctx);
/* Open-loop check entry: one correlate-and-dump epoch per channel with the caller's
 * NCO state (no loop update).  state: per channel {sample offset (0-based), blksize,
 * remCodePhase, codeFreq, remCarrPhase, carrFreq}; sums: [n_ch][18] raw correlator
 * outputs in the order I_E,Q_E,I_P,Q_P,I_L,Q_L, pilot(6), pilot BOC61 (6). */
BDS_API int bds_track_correlate(bds_ctx *ctx, const bds_settings *s, const int8_t *file_bytes,
                                size_t n_bytes, int n_ch, const int32_t *prn,
                                const double *state6, double *sums18);

/* ---- helpers replacing small host functions on the path ------------------------ */
/* Common/calcLoopCoef.m:41-45 */
BDS_API void bds_calc_loop_coef(double lbw, double zeta, double k, double *tau1, double *tau2);
/* Common/calcLoopCoefCarr.m:41-56 */
BDS_API void bds_calc_loop_coef_carr(const bds_settings *s, double *pf3, double *pf2, double *pf1);
/* B1C/include/CalcWeighingFactor.m:43-81 */
BDS_API double bds_calc_weighing_factor(const bds_settings *s);
/* preRun.m:61-76 (B1C applies Doppler aiding to codeFreq, B2a does not) */
BDS_API int bds_pre_run(const bds_settings *s, int max_prn, const double *carrFreq,
                        const double *codePhase, const double *peakMetric, bds_channel *channel);
/* The same allocation as a device kernel (stable descending rank of peakMetric by counting, one wave): replaces the host
 * loop of include/preRun.m:61-76 (either receiver) when acquisition and tracking are chained inside the library; `channel` (host,
 * numberOfChannels entries) receives a copy.  Bit-identical to bds_pre_run; settings.numberOfChannels > 64 (more channels than
 * PRNs: the extra ones stay idle) is served by the host loop. */
BDS_API int bds_pre_run_device(bds_ctx *ctx, const bds_settings *s, int max_prn, const double *carrFreq,
                               const double *codePhase, const double *peakMetric, bds_channel *channel);
/* acquisition -> preRun -> tracking in ONE call (B2a/postProcessing.m:100-123, B1C/postProcessing.m:105-143 without the
 * MATLAB statements between them): bds_acquire on `samples`, bds_pre_run_device on its results, bds_track on the record
 * at `path` with the settings' own tracking variant.  acqResults (as bds_acquire) and the channel table
 * (numberOfChannels entries) are returned beside trackResults. */
BDS_API int bds_acquire_track(bds_ctx *ctx, const bds_settings *s, const int8_t *samples, size_t n_samples,
                              int is_complex, int max_prn, double *carrFreq, double *codePhase,
                              double *peakMetric, int32_t *detected, const char *path,
                              bds_channel *channel, bds_track_out *out);
/* ---- frame synchronisation correlators (the first consumers of trackResults) -------------
 * B1C/include/BCNAV1decoding.m:66-91: bits = sign(Pilot_I_P) (wide-band tracking) or sign(Pilot_Q_P)
 *   (narrow-band), XcorrResult = second half of xcorr(bits, generate2ndCode(PRN)) (1800 chips),
 *   index = find(abs(XcorrResult) >= 1799.5).
 * B2a/include/BCNAV2decoding.m:69-97: bits = sign(I_P), pattern = kron(preamble_bits, secondCode)
 *   (120 taps), index = find(abs(.) > 115).
 * prompt: double[n_ch * n] prompt-correlator series, one row per channel; prn[n_ch].
 * xcorr (optional): int32[n_ch * M], M = max(n, pattern length): the correlation at lags 0..M-1.
 * index (optional): int32[n_ch * cap], 1-based like find(); n_index[n_ch] = number of hits per
 *   channel (may exceed cap).  Returns the total number of hits or <0.
 * bds_sync_pattern: the +-1 pattern itself (1800 or 120 values). */
BDS_API int bds_frame_sync(bds_ctx *ctx, int signal, int n_ch, const int32_t *prn, const double *prompt,
                           int n, int32_t *xcorr, int32_t *index, int32_t *n_index, int cap);
BDS_API int bds_sync_pattern(int signal, int prn, int8_t *out, int cap);

/* B2a/include/unpack_cplx.m: converter of packed records (one byte = two complex samples, 2-bit
 * sign/magnitude I and Q each) into the int8 I/Q pairs of a fileType-2 record.
 * out: int8[4 * n_bytes] = I1, Q1, I2, Q2 per input byte.  The file variant mirrors
 * unpack_cplx(filename_in, filename_out). */
BDS_API int bds_unpack_cplx(bds_ctx *ctx, const uint8_t *in, size_t n_bytes, int8_t *out);
BDS_API int bds_unpack_cplx_file(bds_ctx *ctx, const char *path_in, const char *path_out);

/* The acquisition's resampling branch (B2a/acquisition.m:54-124, B1C/acquisition.m:54-123), taken
 * inside bds_acq_load when samplingFreq > resamplingThreshold && resamplingflag == 1.
 * bds_resample_plan: returns 1 and the sampling rate / IF acquisition() continues with (:103,:119)
 *   plus the fir1 band edges (:66) when the branch is taken, 0 (outputs untouched) otherwise.
 * bds_fir1_bandpass: b = fir1(n_taps-1, [wp1 wp2]) (Hamming window, unit gain at the band centre). */
BDS_API int bds_resample_plan(const bds_settings *s, double *new_fs, double *new_if, double *wp);
BDS_API int bds_fir1_bandpass(int n_taps, double wp1, double wp2, double *b);

#ifdef __cplusplus
}
#endif
#endif /* BDS_MI355X_H */
