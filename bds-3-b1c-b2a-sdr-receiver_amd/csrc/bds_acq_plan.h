// Two-pass plan of the acquisition transforms (included by bds_acq.hip inside namespace bds): the factorisation L = L1 x L2
// of a 5-smooth transform length, its stage radices, the cost model that picks it (choose_lengths), the twiddle / per-lane
// constant tables of the search kernels, and plan_build.  Split out of bds_acq.hip in round 5 (one 2 000-line file).
#pragma once
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

