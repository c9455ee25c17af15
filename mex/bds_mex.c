/*
 * bds_mex.c -- MEX gateway: MATLAB <-> libbds_mi355x.so (C ABI of include/bds_mi355x.h).
 *
 * Build (on a machine that has MATLAB; this image has none, so the file is kept
 * logic-free -- everything it calls is exercised through the same C ABI by the
 * repo's ctypes layer and tests):
 *
 *   mex -R2018a -I../include bds_mex.c -L../bds-3-b1c-b2a-sdr-receiver_amd -lbds_mi355x \
 *       LDFLAGS='$LDFLAGS -Wl,-rpath,/opt/rocm/lib'
 *
 * Usage from the drop-in wrappers in this directory:
 *   [carrFreq, codePhase, peakMetric, detected] = bds_mex('acquire', int8(longSignal), settings, signal, iq)
 *       iq (optional, default false): longSignal holds interleaved I/Q pairs (fileType 2)
 *   out = bds_mex('track', path, channel, settings, signal)        % struct of [nCh x nEpochs] arrays
 *   code = bds_mex('gen_code', signal, kind, prn)
 *   [XcorrResult, index] = bds_mex('frame_sync', signal, PRN, bits)   % one channel: second half of
 *       xcorr(sign(bits), pattern) and find(abs(.) >= 1799.5) (B1C) / find(abs(.) > 115) (B2a)
 * signal: 1 = B1C, 2 = B2a (the reference keeps one directory per receiver).
 * Acquisition runs on ONE GPU unless BDS_MEX_DEVICES=n (n > 1, or 0 = every visible GPU) opts into the multi-device
 * path of the library (bds_multi_create + RCCL all-reduce): that path has only ever run on one-GPU boxes (with the exchange
 * forced, and with two contexts aliased onto one device), so it is not the default of a MATLAB session.
 * No MATLAB exists in the build image: tests/test_mex_syntax.py compiles this file with -fsyntax-only against a header that
 * declares the MEX API, and tests/test_mex_mock.py builds it against a small stand-in for the MEX runtime
 * (tests/mex_stub/mex_mock.c) and runs every command and every error exit through ctypes.
 */
#include <stdlib.h>
#include <string.h>

#include "bds_mi355x.h"
#include "mex.h"

/* One bds_multi for the MATLAB session: one GPU by default, BDS_MEX_DEVICES=n GPUs (0 = all visible).  Acquisition
 * goes through bds_acquire_multi (PRN shards + RCCL all-reduce inside the library); tracking, frame sync and the
 * converters run on device 0 of it (tracking needs no exchange: replicas only). */
static bds_multi *g_multi = NULL;

static void cleanup(void) {
    if (g_multi) bds_multi_destroy(g_multi);
    g_multi = NULL;
}

static bds_multi *multi(void) {
    if (!g_multi) {
        const char *e = getenv("BDS_MEX_DEVICES");
        g_multi = bds_multi_create(e ? atoi(e) : 1, NULL);
        if (!g_multi) mexErrMsgIdAndTxt("bds:create", "%s", bds_multi_last_error(NULL));
        mexAtExit(cleanup);
    }
    return g_multi;
}

static bds_ctx *ctx(void) { return bds_multi_ctx(multi(), 0); }

/* Every field of SURVEY.md Appendix D that the selected receiver's initSettings.m defines is REQUIRED: a settings
 * struct with a misspelt or missing field is an error naming the field, never a silent default. */
static const mxArray *need(const mxArray *s, const char *name) {
    const mxArray *f = mxGetField(s, 0, name);
    if (!f) mexErrMsgIdAndTxt("bds:settings", "settings.%s is missing", name);
    return f;
}
static double num(const mxArray *s, const char *name) {
    const mxArray *f = need(s, name);
    if (!(mxIsNumeric(f) || mxIsLogical(f)) || mxGetNumberOfElements(f) != 1)
        mexErrMsgIdAndTxt("bds:settings", "settings.%s must be a numeric scalar", name);
    return mxGetScalar(f);
}

static void pack_settings(const mxArray *s, int signal, bds_settings *o) {
    const mxArray *lst, *dt;
    char dts[32];
    size_t n, i;
    const double *p;
    if (!mxIsStruct(s)) mexErrMsgIdAndTxt("bds:settings", "settings must be a struct");
    if (signal != BDS_SIGNAL_B1C && signal != BDS_SIGNAL_B2A) mexErrMsgIdAndTxt("bds:settings", "signal must be 1 (B1C) or 2 (B2a)");
    memset(o, 0, sizeof(*o));
    o->signal = signal;
    /* both receivers (B1C/initSettings.m:48-151, B2a/initSettings.m:44-130) */
    o->fileType = (int)num(s, "fileType");
    o->samplingFreq = num(s, "samplingFreq");
    o->IF = num(s, "IF");
    o->codeFreqBasis = num(s, "codeFreqBasis");
    o->carrFreqBasis = num(s, "carrFreqBasis");
    o->codeLength = (int)num(s, "codeLength");
    o->numberOfChannels = (int)num(s, "numberOfChannels");
    o->skipNumberOfBytes = (int64_t)num(s, "skipNumberOfBytes");
    o->msToProcess = num(s, "msToProcess");
    o->acqSearchBand = num(s, "acqSearchBand");
    o->acqStep = num(s, "acqStep");
    o->acqThreshold = num(s, "acqThreshold");
    o->resamplingThreshold = num(s, "resamplingThreshold");
    o->resamplingflag = (int)num(s, "resamplingflag");
    o->pilotTRKflag = (int)num(s, "pilotTRKflag");
    o->intTime = num(s, "intTime");
    o->dllCorrelatorSpacing = num(s, "dllCorrelatorSpacing");
    o->dllDampingRatio = num(s, "dllDampingRatio");
    o->dllNoiseBandwidth = num(s, "dllNoiseBandwidth");
    o->pllNoiseBandwidth = num(s, "pllNoiseBandwidth");
    o->CNoInterval = (int)num(s, "CNoInterval");
    /* settings.dataType: the library reads int8 ('schar') records only (tracking.m:237-238) */
    dt = need(s, "dataType");
    if (!mxIsChar(dt) || mxGetString(dt, dts, sizeof(dts)) || strcmp(dts, "schar"))
        mexErrMsgIdAndTxt("bds:settings", "settings.dataType must be 'schar' (int8 samples)");
    o->dataType = 0;
    if (signal == BDS_SIGNAL_B1C) { /* B1C/initSettings.m:60,70,102 */
        o->acqCohT = num(s, "acqCohT");
        o->pilotACQflag = (int)num(s, "pilotACQflag");
        o->FEBW = num(s, "FEBW");
        o->fineNoncoh = 1;
    } else { /* B2a/initSettings.m:84 */
        o->fineNoncoh = (int)num(s, "fineNoncoh");
        o->acqCohT = 10;
        o->pilotACQflag = 1;
    }
    lst = need(s, "acqSatelliteList");
    if (!mxIsDouble(lst)) mexErrMsgIdAndTxt("bds:settings", "settings.acqSatelliteList must be a double row vector");
    n = mxGetNumberOfElements(lst);
    p = mxGetDoubles(lst);
    if (n > BDS_MAX_PRN) mexErrMsgIdAndTxt("bds:settings", "settings.acqSatelliteList longer than 63");
    o->n_acq = (int)n;
    for (i = 0; i < n; ++i) o->acqSatelliteList[i] = (int)p[i];
}

static void do_acquire(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    bds_settings s;
    int max_prn = 0, i, rc, iq;
    mxArray *det;
    if ((nrhs != 4 && nrhs != 5) || !mxIsInt8(prhs[1]))
        mexErrMsgIdAndTxt("bds:args", "acquire: (int8 longSignal, settings, signal[, iq])");
    iq = nrhs == 5 && mxGetScalar(prhs[4]) != 0;
    pack_settings(prhs[2], (int)mxGetScalar(prhs[3]), &s);
    for (i = 0; i < s.n_acq; ++i)
        if (s.acqSatelliteList[i] > max_prn) max_prn = s.acqSatelliteList[i];
    plhs[0] = mxCreateDoubleMatrix(1, max_prn, mxREAL);
    plhs[1] = mxCreateDoubleMatrix(1, max_prn, mxREAL);
    plhs[2] = mxCreateDoubleMatrix(1, max_prn, mxREAL);
    det = mxCreateNumericMatrix(1, max_prn, mxINT32_CLASS, mxREAL);
    {
        bds_acq_job job;
        memset(&job, 0, sizeof(job));
        job.settings = &s;
        job.samples = (const int8_t *)mxGetInt8s(prhs[1]);
        job.n_samples = mxGetNumberOfElements(prhs[1]) / (iq ? 2 : 1);
        job.is_complex = iq;
        job.max_prn = max_prn;
        job.carrFreq = mxGetDoubles(plhs[0]);
        job.codePhase = mxGetDoubles(plhs[1]);
        job.peakMetric = mxGetDoubles(plhs[2]);
        job.detected = (int32_t *)mxGetInt32s(det);
        rc = bds_acquire_multi(multi(), 1, &job); /* all GPUs of the session: PRN shards + one RCCL all-reduce */
    }
    if (rc) mexErrMsgIdAndTxt("bds:acquire", "%s", bds_multi_last_error(g_multi));
    if (nlhs > 3)
        plhs[3] = det;
    else
        mxDestroyArray(det);
}

static double *out_field(mxArray *st, const char *name, int rows, int cols) {
    mxArray *a = mxCreateDoubleMatrix(rows, cols, mxREAL); /* column-major: [epoch x channel] */
    mxAddField(st, name);
    mxSetField(st, 0, name, a);
    return mxGetDoubles(a);
}

static void do_track(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    bds_settings s;
    bds_track_out o;
    bds_channel *ch;
    char path[4096];
    int n_ch, n_ep, n_cno, c, rc, wb, pilot;
    mxArray *st, *comp, *stat;
    (void)nlhs;
    if (nrhs != 5) mexErrMsgIdAndTxt("bds:args", "track: (path, channel, settings, signal)");
    mxGetString(prhs[1], path, sizeof(path));
    pack_settings(prhs[3], (int)mxGetScalar(prhs[4]), &s);
    n_ch = (int)mxGetNumberOfElements(prhs[2]);
    ch = (bds_channel *)mxCalloc((size_t)n_ch, sizeof(bds_channel));
    for (c = 0; c < n_ch; ++c) {
        const mxArray *stc = mxGetField(prhs[2], c, "status");
        ch[c].PRN = (int)mxGetScalar(mxGetField(prhs[2], c, "PRN"));
        ch[c].acquiredFreq = mxGetScalar(mxGetField(prhs[2], c, "acquiredFreq"));
        ch[c].codePhase = mxGetScalar(mxGetField(prhs[2], c, "codePhase"));
        ch[c].codeFreq = mxGetScalar(mxGetField(prhs[2], c, "codeFreq"));
        ch[c].status = stc ? (int)mxGetChars(stc)[0] : '-';
    }
    /* epochs: msToProcess (B2a/tracking.m:100) or round(msToProcess/1000/intTime) (WB_tracking.m:56) */
    n_ep = s.signal == BDS_SIGNAL_B2A ? (int)s.msToProcess : (int)(s.msToProcess / 1000 / s.intTime + 0.5);
    n_cno = n_ep / s.CNoInterval;
    wb = s.signal == BDS_SIGNAL_B1C && s.pilotTRKflag == 2;
    pilot = wb || s.pilotTRKflag == 1;
    memset(&o, 0, sizeof(o));
    o.n_ch = n_ch;
    o.n_epochs = n_ep;
    o.n_cno = n_cno;
    st = mxCreateStructMatrix(1, 1, 0, NULL);
#define F(name) o.name = out_field(st, #name, n_ep, n_ch)
    F(absoluteSample); F(codeFreq); F(carrFreq); F(I_P); F(I_E); F(I_L); F(Q_E); F(Q_P); F(Q_L);
    if (pilot) { F(Pilot_I_P); F(Pilot_Q_P); }
    if (wb) { F(Pilot_I_E); F(Pilot_I_L); F(Pilot_Q_E); F(Pilot_Q_L); }
    F(dllDiscr); F(dllDiscrFilt); F(pllDiscr); F(pllDiscrFilt); F(remCodePhase); F(remCarrPhase);
#undef F
#define G(name) o.name = out_field(st, #name, n_cno, n_ch)
    G(DataCNo); G(DataPLD);
    if (pilot) { G(PilotCNo); G(PilotPLD); G(SigCNo); }
#undef G
    comp = mxCreateNumericMatrix(1, n_ch, mxINT32_CLASS, mxREAL);
    stat = mxCreateNumericMatrix(1, n_ch, mxINT32_CLASS, mxREAL);
    o.completed = (int32_t *)mxGetInt32s(comp);
    o.status = (int32_t *)mxGetInt32s(stat);
    mxAddField(st, "completed");
    mxSetField(st, 0, "completed", comp);
    mxAddField(st, "status");
    mxSetField(st, 0, "status", stat);
    /* the C arrays are channel-major [n_ch][n_epochs] == MATLAB column-major [n_epochs x n_ch] */
    rc = bds_track(ctx(), &s, path, n_ch, ch, &o);
    mxFree(ch);
    if (rc) mexErrMsgIdAndTxt("bds:track", "%s", bds_last_error(ctx()));
    plhs[0] = st;
}

static void do_gen_code(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    int8_t buf[122760];
    int n, i;
    double *p;
    (void)nlhs;
    if (nrhs != 4) mexErrMsgIdAndTxt("bds:args", "gen_code: (signal, kind, prn)");
    n = bds_gen_code((int)mxGetScalar(prhs[1]), (int)mxGetScalar(prhs[2]), (int)mxGetScalar(prhs[3]), buf, 122760);
    if (n < 0) mexErrMsgIdAndTxt("bds:gen_code", "bad signal/kind/prn");
    plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);
    p = mxGetDoubles(plhs[0]);
    for (i = 0; i < n; ++i) p[i] = buf[i];
}

/* BCNAV1decoding.m:75-91 / BCNAV2decoding.m:84-97 for one channel */
static void do_frame_sync(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    int signal, n, m, M, i, total;
    int32_t prn, cnt = 0, *xc, *idx;
    double *p;
    if (nrhs != 4 || !mxIsDouble(prhs[3])) mexErrMsgIdAndTxt("bds:args", "frame_sync: (signal, PRN, bits)");
    signal = (int)mxGetScalar(prhs[1]);
    prn = (int32_t)mxGetScalar(prhs[2]);
    n = (int)mxGetNumberOfElements(prhs[3]);
    m = signal == BDS_SIGNAL_B1C ? 1800 : 120;
    M = n > m ? n : m;
    xc = (int32_t *)mxCalloc((size_t)M, sizeof(int32_t));
    idx = (int32_t *)mxCalloc((size_t)M, sizeof(int32_t));
    total = bds_frame_sync(ctx(), signal, 1, &prn, mxGetDoubles(prhs[3]), n, xc, idx, &cnt, M);
    if (total < 0) mexErrMsgIdAndTxt("bds:frame_sync", "%s", bds_last_error(ctx()));
    plhs[0] = mxCreateDoubleMatrix(1, M, mxREAL);
    p = mxGetDoubles(plhs[0]);
    for (i = 0; i < M; ++i) p[i] = xc[i];
    if (nlhs > 1) {
        plhs[1] = mxCreateDoubleMatrix(cnt, 1, mxREAL); /* index = find(...)' is a column */
        p = mxGetDoubles(plhs[1]);
        for (i = 0; i < cnt; ++i) p[i] = idx[i];
    }
    mxFree(xc);
    mxFree(idx);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[32];
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof(cmd))) mexErrMsgIdAndTxt("bds:args", "first argument: command string");
    if (!strcmp(cmd, "acquire"))
        do_acquire(nlhs, plhs, nrhs, prhs);
    else if (!strcmp(cmd, "track"))
        do_track(nlhs, plhs, nrhs, prhs);
    else if (!strcmp(cmd, "gen_code"))
        do_gen_code(nlhs, plhs, nrhs, prhs);
    else if (!strcmp(cmd, "frame_sync"))
        do_frame_sync(nlhs, plhs, nrhs, prhs);
    else
        mexErrMsgIdAndTxt("bds:args", "unknown command %s", cmd);
}
