// What bds_acq_run decides on the sieve's output (included by bds_acq.hip inside namespace bds { namespace {, after AcqRun):
//   host path    collect() -> refine() -> metric_b1c() / second_peak_b2a() -> fine_search()
//   device chain refine_device(): the same decisions as one chain of launches with a single download (bds_acq_refine.h),
//                the host forming the reported numbers from the winners' sums
// Split out of bds_acq.hip in round 5 (one 2 000-line file).
#pragma once
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
    if ((rc = run_jobs(ctx, a, *s, jobs, jout, ncomp))) return rc;
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
        const double mean = (a.prefix(a.X) - a.prefix(0)) / (double)a.X;
        const double mean_q = a.cplx ? (a.prefix(a.X, 1) - a.prefix(0, 1)) / (double)a.X : 0.0;
        long double acc = 0;
        for (long i = 0; i < a.X; ++i) {
            const double d = a.sample_re(i) - mean;
            const double dq = a.cplx ? a.sample_im(i) - mean_q : 0.0;
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
    if ((rc = run_jobs(ctx, a, *s, jobs, jout, ncomp))) return rc;
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
            const double mean = (a.prefix(r.codePhase - 1 + a.spc) - a.prefix(r.codePhase - 1)) / (double)a.spc;  // :254
            const double mean_q = a.cplx ? (a.prefix(r.codePhase - 1 + a.spc, 1) - a.prefix(r.codePhase - 1, 1)) / (double)a.spc : 0.0;
            for (int kf = 0; kf < nfine; ++kf) fine_frq[pi].push_back(fb - s->acqStep + 25.0 * kf);  // :282-283
            // jobs of one PRN: [chunk of up to kCorrFreqs frequencies][component] (the components of a chunk are summed in one pass)
            for (int k0 = 0; k0 < nfine; k0 += kCorrFreqs)
                for (int comp = 0; comp < ncomp; ++comp) {
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
            // jobs of one PRN: [segment][chunk of up to kCorrFreqs frequencies][component]
            for (int seg = 0; seg < s->fineNoncoh; ++seg)
                for (int k0 = 0; k0 < nfine; k0 += kCorrFreqs)
                    for (int comp = 0; comp < 2; ++comp) {
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
    if ((rc = run_jobs(ctx, a, *s, jobs, jout, a.signal == BDS_SIGNAL_B1C ? ncomp : 2, true))) return rc;
    const int nchunk = (nfine + kCorrFreqs - 1) / kCorrFreqs;
    size_t job0 = 0;  // first job of the PRN
    for (int pi = 0; pi < P; ++pi) {
        if (fine_of[pi] < 0) continue;
        // sum of frequency kf of job group (seg, comp): jobs are laid out [seg][chunk][comp]
        auto at = [&](int seg, int comp, int ncomp_, int kf) {
            const size_t j = job0 + ((size_t)seg * nchunk + kf / kCorrFreqs) * ncomp_ + comp;
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
    {  // the fine-frequency pick keeps its sums in LDS: thousands of fine frequencies (acqStep / 25 large) go to the host path BEFORE
       // anything of the device chain is enqueued (round 6, ADVICE r5: the test stood behind the whole job chain, which then ran twice)
        const bool b1c = a.signal == BDS_SIGNAL_B1C;
        const size_t nfine = b1c ? (size_t)m_round(s->acqStep / 25) * 2 + 1 : (size_t)m_round(s->acqStep / 25) + 1;  // B1C :267, B2a :265
        if (sizeof(double) * ((size_t)(b1c ? ncomp : 2 * s->fineNoncoh) * nfine + nfine) > 60000) return false;
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
    const char *tabs_before = a.d_ref_tabs;
    if ((rc = ensure(ctx, &a.d_ref_tabs, &a.ref_tabs_cap, (sizeof(long) + sizeof(int)) * (size_t)P + 64))) return rc;
    long *d_cs_of = (long *)a.d_ref_tabs;
    int *d_prn_of = (int *)(d_cs_of + P);
    std::vector<long> h_cs(P);
    for (int pi = 0; pi < P; ++pi) h_cs[pi] = (long)a.cs_slot[prns[pi]] * ncomp * pl.L;
    if (a.d_ref_tabs != tabs_before || a.ref_tabs_cs != h_cs || a.ref_tabs_prn != prns) {  // (the same PRN list call after call: no upload)
        a.ref_tabs_cs.clear(), a.ref_tabs_prn.clear();
        BDS_HIP(ctx, hipMemcpyAsync(d_cs_of, h_cs.data(), sizeof(long) * P, hipMemcpyHostToDevice, sm));
        BDS_HIP(ctx, hipMemcpyAsync(d_prn_of, prns.data(), sizeof(int) * P, hipMemcpyHostToDevice, sm));
        BDS_HIP(ctx, hipStreamSynchronize(sm));
        a.ref_tabs_cs = h_cs, a.ref_tabs_prn = prns;
    }
    const unsigned pb = (unsigned)((P + 63) / 64);

    // ---- coarse refinement: thresholds -> candidates in the band -> f64 sums -> per-PRN maximum ----------------
    hipLaunchKernelGGL(k_ref_thr<false>, dim3(P), dim3(64), 0, sm, (const unsigned long long *)a.d_cellmax, rp, a.d_ref_prn, a.d_ref_g,
                       (const int *)a.d_extra_count);
    hipLaunchKernelGGL(k_ref_compact<false>, dim3(256), dim3(256), 0, sm, (const Extra *)a.d_extra, (const int *)a.d_extra_count, rp,
                       (const RefPrn *)a.d_ref_prn, (const int *)d_prn_of, (const int4 *)nullptr, a.d_ref_cand, a.d_jobs, a.d_ref_g);
    launch_corr<1>(sm, dim3(1024, kCorrSlices), a.sview(), ncomp, a.N, (const int8_t *)a.d_codes, a.code_stride, 1.0 / a.fs,
                   (const CorrJob *)a.d_jobs, a.d_jobout, (const int *)&a.d_ref_g->ncand, kRefCandCap);
    hipLaunchKernelGGL(k_ref_pick<false>, dim3(P), dim3(256), 0, sm, (const RefCand *)a.d_ref_cand, (const double2 *)a.d_jobout, kCorrSlices,
                       rp, a.d_ref_prn, a.d_ref_g);
    BDS_HIP(ctx, hipGetLastError());

    // ---- B2a: second peak of the winning bin, outside +-2 chips and within +-1 code (acquisition.m:224-249) -------
    if (!b1c) {
        if ((rc = ensure(ctx, &a.d_extra2, &a.extra2_cap, (size_t)kExtra2Cap))) return rc;
        // The main search of a small grid ran ALL P x D cells in one launch pair (multiprn, cfg2: 1 638 cells, 4.3 GB): the rows of every
        // winning cell still lie in the inter-pass buffer, so the second-peak pass needs no row pass of its own -- its column pass
        // reads cell pi D + b (round 5: 0.11 ms of cfg2's 0.5 ms refinement; BDS_ACQ_NO_BWREUSE of the hooks build switches back)
        const bool reuse_bw = multiprn && n_pairs_total == 1 && pl.small && !tune.no_bwreuse;
        const size_t nb_ = 2 * sizeof(int) * P + sizeof(long) * P + sizeof(int4) * P + 64;
        if ((rc = ensure(ctx, &a.d_cells, &a.cells_cap, nb_))) return rc;
        int4 *d_rng = (int4 *)a.d_cells;  // 16-byte aligned first
        long *d_cs = (long *)(d_rng + P);
        int *d_bin = (int *)(d_cs + P);
        int *d_src = d_bin + P;
        hipLaunchKernelGGL(k_ref_second_setup, dim3(pb), dim3(64), 0, sm, rp, a.d_ref_prn, (const long *)d_cs_of, d_rng, d_cs, d_bin,
                           reuse_bw ? d_src : (int *)nullptr, a.d_ref_g);
        const SieveOut so_keep = so;
        so.recs = nullptr;
        so.extra = a.d_extra2, so.extra_count = a.d_extra2_count, so.extra_cap = kExtra2Cap;
        so.cellmax = a.d_cellmax2, so.lb = a.d_lb2, so.lb_div = 1;
        CellList cl{d_bin, d_cs, d_rng};
        if (reuse_bw) cl.src = d_src;
        launch_list(P, nullptr, cl, 0, nullptr);
        so = so_keep;
        RefParams rp2 = rp;
        rp2.extra_cap = kExtra2Cap;
        RefCand *cand2 = a.d_ref_cand + kRefCandCap;
        hipLaunchKernelGGL(k_ref_thr<true>, dim3(P), dim3(64), 0, sm, (const unsigned long long *)a.d_cellmax2, rp2, a.d_ref_prn, a.d_ref_g,
                           (const int *)a.d_extra2_count);
        hipLaunchKernelGGL(k_ref_compact<true>, dim3(64), dim3(256), 0, sm, (const Extra *)a.d_extra2, (const int *)a.d_extra2_count, rp2,
                           (const RefPrn *)a.d_ref_prn, (const int *)d_prn_of, (const int4 *)d_rng, cand2, a.d_jobs, a.d_ref_g);
        launch_corr<1>(sm, dim3(1024, kCorrSlices), a.sview(), ncomp, a.N, (const int8_t *)a.d_codes, a.code_stride, 1.0 / a.fs,
                       (const CorrJob *)a.d_jobs, a.d_jobout, (const int *)&a.d_ref_g->ncand2, kRefCandCap);
        hipLaunchKernelGGL(k_ref_pick<true>, dim3(P), dim3(256), 0, sm, (const RefCand *)cand2, (const double2 *)a.d_jobout, kCorrSlices, rp2,
                           a.d_ref_prn, a.d_ref_g);
        BDS_HIP(ctx, hipGetLastError());
    }

    // ---- threshold + fine-Doppler search --------------------------------------------------------------------
    hipLaunchKernelGGL(k_ref_fine_jobs, dim3(P), dim3(64), 0, sm, rp, a.d_ref_prn, (const int *)d_prn_of, a.sview(), (const double *)a.d_prefix_c,
                       (const double *)a.d_prefix_cq, a.d_jobs, a.d_ref_g);
    const int fine_nc = b1c ? ncomp : 2;  // the components of a (segment, chunk) are adjacent jobs, summed in one pass
    // (the job list is compact -- only detected PRNs have jobs, their count lives on the device -- and a fixed grid walks it: a launch
    //  over every PRN's slots started 22 680 workgroups at cfg2 to find 3 600 with work)
    const int fine_units = P * fine_per / fine_nc;
    launch_corr<kCorrFreqs>(sm, dim3((unsigned)std::min(fine_units, 1024), kCorrSlices), a.sview(), fine_nc, a.N, (const int8_t *)a.d_codes,
                            a.code_stride, 1.0 / a.fs, (const CorrJob *)a.d_jobs, a.d_jobout, (const int *)&a.d_ref_g->nfine_units, fine_units);
    const size_t pick_lds = sizeof(double) * ((size_t)(b1c ? ncomp : 2 * s->fineNoncoh) * rp.nfine + rp.nfine);
    if (pick_lds > 60000) return kHostRefine;  // (cannot happen: device_refine_ok() sends such settings to the host path up front)
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

