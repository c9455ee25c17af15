// Device-side refinement of the search's candidates (round 5): the sieve's candidate list -> f64 coherent sums -> per-PRN
// peak / bin / code phase (first row / first column on ties, as MATLAB max: B1C/acquisition.m:218-221, B2a/acquisition.m:213-222)
// -> B2a second peak (B2a/acquisition.m:224-249) -> threshold -> fine-Doppler search (B1C :244-307, B2a :255-336), as ONE chain of
// launches on the search's stream with a single download at its end.  Rounds 1-4 built the f64 jobs on the host from
// downloaded lists (three to six stream synchronisations per call, std::set job building: 0.65 ms of cfg2's 1.25 ms
// refinement); the host path is still there (AcqRun::refine ...) for the tile-record column pass, the +-n neighbour ring,
// resampled (non-integer) blocks, and as the reference the test-hooks build compares this chain with (BDS_ACQ_HOSTREFINE=1).
//
// What the device decides and what the host computes: the device picks -- which candidate is the maximum, which lag holds the
// second peak, whether the metric passes the threshold, which fine frequency wins -- and hands back the winners' coherent
// sums; the host forms the reported numbers from those sums with the very expressions of the host path (combine(), hypot of
// the C library), so acqResults are bit-identical between the two paths.
#pragma once

#include "bds_acq_kernels.h"

namespace bds {

struct RefCand {
    int pi;   // PRN index within the run
    int b;    // 0-based bin
    int lag;  // 0-based
};

enum : int {
    kRefNonFinite = 1,    // a row maximum of the sieve is not finite
    kRefSelfCheck = 2,    // f64 peak vs sieve maximum beyond kDelta / 2
    kRefEmptyRange = 4,   // B2a: empty second-peak range (acquisition.m:248 would fail)
    kRefFineRange = 8,    // fine-search block outside longSignal
    kRefCandOverflow = 16 // more candidates than the job buffers hold: the host path takes over
};

struct RefPrn {  // per PRN of the run
    double2 v[2];    // coherent sums of the winner, per component
    double2 v2[2];   // B2a: ... of the second peak
    double best;     // device-side value of the winner (selection only; the host recomputes the reported number from v)
    double second;   // B2a: device-side value of the second peak
    double mean, mean_q;  // B1C: DC of the fine-search block (acquisition.m:254)
    long codePhase;  // 1-based, after the B1C end-of-block adjustment (:239-241)
    int b, lag;      // winner, 0-based
    int ncand, nsecond;
    int detected, kbest;
    int flags;
    float thr, max_of;    // sieve threshold and sieve maximum of the PRN
    float thr2, max2;     // ... of the second-peak pass
    int fine_job0;        // first of this PRN's fine-search jobs in the (compact) job list
};

struct RefGlobal {
    int ncand;           // candidates of the coarse refinement
    int nfine_units;     // fine-search units (jobs / components) of the detected PRNs: the list holds no place-holders
    int ncand2, pad1;    // ... of the B2a second peak
    int flags;
    int n_extra, n_extra2;  // list lengths as the column pass left them (may exceed the capacity: overflow)
    int pad;
};

struct RefParams {
    int P, D, ncomp, signal;
    int half;            // fp16 storage: run the self-check
    int cand_cap;        // candidates d_cand / d_jobs hold
    int extra_cap;
    double kDelta;
    double f0, step;     // bin_freq(b) = f0 + step * b
    long X, N, spc, n_samples;
    double threshold;
    double sigPower;     // B1C normaliser
    long s2c;            // B2a: samples2CodeChip
    int fineNoncoh;      // B2a
    int nfine, nchunk;
    int cplx;
};

__device__ __forceinline__ double ref_bin_freq(const RefParams &p, int b) {
    return __dadd_rn(p.f0, __dmul_rn(p.step, (double)b));  // (no contraction: the host forms f0 + step * b with two roundings)
}
__device__ __forceinline__ double ref_cabs(double2 v) { return hypot(v.x, v.y); }
__device__ __forceinline__ double ref_combine(const RefParams &p, const double2 *v) {
    if (p.signal == BDS_SIGNAL_B2A) return __dadd_rn(ref_cabs(v[0]), ref_cabs(v[1]));
    if (p.ncomp == 1) return ref_cabs(v[0]);
    return __ddiv_rn(__dadd_rn(__dmul_rn(ref_cabs(v[0]), sqrt(11.0)), __dmul_rn(ref_cabs(v[1]), sqrt(29.0))), sqrt(40.0));
}
__device__ __forceinline__ void ref_unpack_cell(unsigned long long pk, float *v, int *lag) {
    if (pk == 0) {
        *v = -1.f, *lag = -1;
        return;
    }
    *v = __uint_as_float((uint32_t)(pk >> 32));
    *lag = (int)~(uint32_t)(pk & 0xffffffffu);
}

// per PRN: sieve maximum over its cells and the threshold of the tolerance band.  grid P, 64 threads.
// SECOND: one cell per PRN (the second-peak pass of the 80 x 4096 plan).
template <bool SECOND>
__global__ __launch_bounds__(64) void k_ref_thr(const unsigned long long *__restrict__ cellmax, RefParams p, RefPrn *__restrict__ prn,
                                                RefGlobal *__restrict__ g, const int *__restrict__ extra_count) {
    const int pi = blockIdx.x, lane = threadIdx.x;
    float M = -1.f;
    bool bad = false;
    const int nc = SECOND ? 1 : p.D;
    for (int b = lane; b < nc; b += 64) {
        float v;
        int lag;
        ref_unpack_cell(cellmax[(size_t)pi * nc + b], &v, &lag);
        bad = bad || !isfinite(v);
        M = fmaxf(M, v);  // (NaN: flagged above)
    }
    for (int o = 32; o > 0; o >>= 1) {
        M = fmaxf(M, __shfl_xor(M, o));
        bad = bad || __shfl_xor((int)bad, o);
    }
    if (lane == 0) {
        const float thr = (float)((1.0 - p.kDelta) * (double)M);
        if (SECOND) {
            prn[pi].thr2 = thr, prn[pi].max2 = M;
        } else {
            prn[pi].thr = thr, prn[pi].max_of = M;
            prn[pi].flags = bad ? kRefNonFinite : 0;
            if (bad) atomicOr(&g->flags, kRefNonFinite);
        }
        if (pi == 0) {
            if (SECOND)
                g->n_extra2 = *extra_count;
            else
                g->n_extra = *extra_count;
        }
    }
}

// candidate list -> candidates inside the tolerance band -> their f64 jobs.  Fixed grid, grid-stride over the list.
// SECOND: the list of the second-peak pass (cell = PRN index; the bin is the winner's; lags outside the two ranges are dropped).
template <bool SECOND>
__global__ __launch_bounds__(256) void k_ref_compact(const Extra *__restrict__ extra, const int *__restrict__ extra_count, RefParams p,
                                                     const RefPrn *__restrict__ prn, const int *__restrict__ prn_of, const int4 *__restrict__ rng,
                                                     RefCand *__restrict__ cand, CorrJob *__restrict__ jobs, RefGlobal *__restrict__ g) {
    const int n = min(*extra_count, p.extra_cap);
    int *const counter = SECOND ? &g->ncand2 : &g->ncand;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Extra e = extra[i];
        const int pi = SECOND ? e.cell : e.cell / p.D;
        if (pi < 0 || pi >= p.P || e.lag < 0) continue;
        const float thr = SECOND ? prn[pi].thr2 : prn[pi].thr;
        if (e.v < thr) continue;
        int b;
        if (SECOND) {
            const int4 r = rng[pi];
            if (!((e.lag >= r.x && e.lag <= r.y) || (e.lag >= r.z && e.lag <= r.w))) continue;
            b = prn[pi].b;
        } else {
            b = e.cell % p.D;
        }
        const int k = atomicAdd(counter, 1);
        if (k >= p.cand_cap) continue;  // (counted all the same: the host sees the overflow and takes the host path)
        cand[k] = RefCand{pi, b, e.lag};
        for (int comp = 0; comp < p.ncomp; ++comp) {
            CorrJob j{};
            j.start = e.lag;
            j.len = p.X;
            j.freq = ref_bin_freq(p, b);
            j.slot = (prn_of[pi] - 1) * 2 + comp;
            j.circ = 1;
            jobs[(size_t)k * p.ncomp + comp] = j;
        }
    }
}

// per PRN: the maximum over its candidates' f64 values; ties: first row (bin), then first column (lag).  grid P, 256 threads.
template <bool SECOND>
__global__ __launch_bounds__(256) void k_ref_pick(const RefCand *__restrict__ cand, const double2 *__restrict__ jobout, int slices, RefParams p,
                                                  RefPrn *__restrict__ prn, RefGlobal *__restrict__ g) {
    const int pi = blockIdx.x, tid = threadIdx.x;
    const int n_all = SECOND ? g->ncand2 : g->ncand;
    const int n = min(n_all, p.cand_cap);
    if (pi == 0 && tid == 0 && n_all > p.cand_cap) atomicOr(&g->flags, kRefCandOverflow);  // the host path takes over
    double best = -1.0;
    int bb = 0, bl = 0, bi = -1, cnt = 0;
    auto sums = [&](int i, double2 *v) {
        for (int c = 0; c < p.ncomp; ++c) {
            double2 acc = make_double2(0.0, 0.0);  // slices in order, as the host adds them
            for (int k = 0; k < slices; ++k) {
                const double2 t = jobout[((size_t)i * p.ncomp + c) * slices + k];
                acc.x = __dadd_rn(acc.x, t.x), acc.y = __dadd_rn(acc.y, t.y);
            }
            v[c] = acc;
        }
    };
    for (int i = tid; i < n; i += 256) {
        const RefCand c = cand[i];
        if (c.pi != pi) continue;
        ++cnt;
        double2 v[2];
        sums(i, v);
        const double val = ref_combine(p, v);
        if (val > best || (val == best && (c.b < bb || (c.b == bb && c.lag < bl)))) best = val, bb = c.b, bl = c.lag, bi = i;
    }
    __shared__ double s_v[256];
    __shared__ int s_b[256], s_l[256], s_i[256], s_n[256];
    s_v[tid] = best, s_b[tid] = bb, s_l[tid] = bl, s_i[tid] = bi, s_n[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const double v2 = s_v[tid + s];
            const int b2 = s_b[tid + s], l2 = s_l[tid + s];
            s_n[tid] += s_n[tid + s];
            if (s_i[tid + s] >= 0 &&
                (s_i[tid] < 0 || v2 > s_v[tid] || (v2 == s_v[tid] && (b2 < s_b[tid] || (b2 == s_b[tid] && l2 < s_l[tid])))))
                s_v[tid] = v2, s_b[tid] = b2, s_l[tid] = l2, s_i[tid] = s_i[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        RefPrn &r = prn[pi];
        double2 v[2] = {make_double2(0, 0), make_double2(0, 0)};
        if (s_i[0] >= 0) sums(s_i[0], v);
        if (SECOND) {
            r.v2[0] = v[0], r.v2[1] = v[1];
            r.second = s_i[0] >= 0 ? s_v[0] : -1.0;
            r.nsecond = s_n[0];
        } else {
            r.v[0] = v[0], r.v[1] = v[1];
            r.best = s_i[0] >= 0 ? s_v[0] : -1.0;
            r.b = s_i[0] >= 0 ? s_b[0] : 0;
            r.lag = s_i[0] >= 0 ? s_l[0] : 0;
            r.ncand = s_n[0];
            r.codePhase = (long)r.lag + 1;
            // the sieve's maximum must agree with the f64 value to well inside the tolerance band it was searched with
            if (p.half && s_n[0] > 0 && fabs(r.best - (double)r.max_of) > 0.5 * p.kDelta * r.best) {
                r.flags |= kRefSelfCheck;
                atomicOr(&g->flags, kRefSelfCheck);
            }
        }
    }
}

// B2a: the (PRN, winning bin) cells of the second-peak pass and their two lag ranges (acquisition.m:224-249).  One thread per PRN.
__global__ void k_ref_second_setup(RefParams p, RefPrn *__restrict__ prn, const long *__restrict__ cs_of, int4 *__restrict__ rng,
                                   long *__restrict__ cs, int *__restrict__ bin, int *__restrict__ src, RefGlobal *__restrict__ g) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= p.P) return;
    const long cp = prn[pi].codePhase;
    const long e1 = cp - p.s2c, e2 = cp + p.s2c, e3 = cp - p.spc + p.s2c, e4 = cp + p.spc - p.s2c;
    long lo1 = 1, hi1 = 0, lo2 = 1, hi2 = 0;  // 1-based inclusive, empty when lo > hi
    if (e1 >= 1) lo1 = e3 > 1 ? e3 : 1, hi1 = e1;
    if (e2 < p.N) lo2 = e2, hi2 = e4 < p.N ? e4 : p.N;
    if (hi1 < lo1 && hi2 < lo2) {
        prn[pi].flags |= kRefEmptyRange;
        atomicOr(&g->flags, kRefEmptyRange);
    }
    rng[pi] = make_int4((int)(lo1 - 1), (int)(hi1 - 1), (int)(lo2 - 1), (int)(hi2 - 1));  // 0-based
    cs[pi] = cs_of[pi];
    bin[pi] = prn[pi].b;
    // (optional) where the main search left this cell's rows in the inter-pass buffer: cell pi D + b of its one launch pair
    if (src) src[pi] = pi * p.D + prn[pi].b;
}

// threshold decision and the jobs of the fine-Doppler search.  A detected PRN appends its jobs
//   B1C [chunk][comp], B2a [segment][chunk][comp]   (chunk = up to kCorrFreqs frequencies 25 Hz apart; the components of a
//   chunk are adjacent: k_corr sums them in one pass)
// to a compact list (RefGlobal::nfine_units counts its units, RefPrn::fine_job0 is the PRN's first job; the order of the PRNs in
// the list follows the atomic, a job's sums do not depend on its place); a PRN below the threshold lists nothing.  One
// workgroup (64 threads) per PRN: thread 0 decides, the DC of the B1C block is summed by all lanes, every thread writes its
// share of the jobs.
__global__ __launch_bounds__(64) void k_ref_fine_jobs(RefParams p, RefPrn *__restrict__ prn, const int *__restrict__ prn_of, SampleView sig,
                                                      const double *__restrict__ prefix_c, const double *__restrict__ prefix_cq,
                                                      CorrJob *__restrict__ jobs, RefGlobal *__restrict__ g) {
    const int pi = blockIdx.x, lane = threadIdx.x;
    RefPrn &r = prn[pi];
    const bool b1c = p.signal == BDS_SIGNAL_B1C;
    const int nseg = b1c ? 1 : p.fineNoncoh, ncp = b1c ? p.ncomp : 2;
    const int per = nseg * ncp * p.nchunk;
    __shared__ long s_cp;
    __shared__ int s_ok, s_job0;
    if (lane == 0) {
        long cp = r.codePhase;
        if (b1c && cp + p.spc - 1 > p.n_samples) cp -= p.spc;  // B1C :239-241
        const double denom = b1c ? p.sigPower : r.second;
        const double metric = __ddiv_rn(r.best, denom);
        const int det = metric > p.threshold ? 1 : 0;
        bool ok = det != 0;
        int fl = 0;
        if (ok) {
            const long blk = b1c ? p.spc : (long)p.fineNoncoh * p.spc;
            if ((b1c && cp < 1) || cp - 1 + blk > p.n_samples) fl = kRefFineRange, ok = false;
        }
        r.codePhase = cp, r.detected = det, r.kbest = 0;
        if (fl) {
            r.flags |= fl;
            atomicOr(&g->flags, fl);
        }
        // only a detected PRN gets jobs: the list is compact (its order follows the atomic, a job's sums do not depend on its place)
        const int job0 = ok ? atomicAdd(&g->nfine_units, per / ncp) * ncp : 0;
        r.fine_job0 = job0;
        s_cp = cp, s_ok = ok ? 1 : 0, s_job0 = job0;
    }
    __syncthreads();
    const long cp = s_cp;
    const bool ok = s_ok != 0;
    if (!ok) return;  // (workgroup-uniform)
    CorrJob *const mine = jobs + s_job0;
    double mean = 0.0, mean_q = 0.0;
    if (ok && b1c) {
        // DC of the block codePhase .. codePhase + spc - 1: exact integer sums (int8 data) from the coarse prefix table
        // (every 256th sample) and the samples between; the host path takes the same integers from its full prefix array
        // (any order of adding them gives the same f64: they are integers below 2^53)
        const long a0 = cp - 1, a1 = a0 + p.spc;
        double d = 0.0, dq = 0.0;  // prefix(a1) - prefix(a0), this lane's share of the samples in between
        for (long m = ((a1 >> 8) << 8) + lane; m < a1; m += 64) {
            const double2 x = sig.load(m);
            d += x.x, dq += x.y;
        }
        for (long m = ((a0 >> 8) << 8) + lane; m < a0; m += 64) {
            const double2 x = sig.load(m);
            d -= x.x, dq -= x.y;
        }
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o), dq += __shfl_xor(dq, o);
        d += prefix_c[a1 >> 8] - prefix_c[a0 >> 8];
        mean = __ddiv_rn(d, (double)p.spc);
        if (p.cplx) {
            dq += prefix_cq[a1 >> 8] - prefix_cq[a0 >> 8];
            mean_q = __ddiv_rn(dq, (double)p.spc);
        }
    }
    if (lane == 0) r.mean = mean, r.mean_q = mean_q;
    const double fb = ref_bin_freq(p, prn[pi].b);
    const double f_lo = b1c ? __dsub_rn(fb, p.step) : __dsub_rn(fb, __ddiv_rn(p.step, 2.0));  // B1C :282-283, B2a :300-301
    for (int i = lane; i < per; i += 64) {
        const int comp = i % ncp, ch = (i / ncp) % p.nchunk, seg = i / (p.nchunk * ncp);
        CorrJob j{};
        j.start = cp - 1 + (long)seg * p.spc;
        j.len = p.spc;
        j.code_k0 = b1c ? 0 : (long)seg * p.spc;
        j.mean = mean;
        j.mean_q = mean_q;
        j.slot = (prn_of[pi] - 1) * 2 + comp;
        j.circ = 0;
        j.mode = b1c ? 0 : 1;
        const int k0 = ch * kCorrFreqs;
        j.nf = ok ? min(kCorrFreqs, p.nfine - k0) : 0;
        for (int f = 0; f < kCorrFreqs; ++f) j.fr[f] = f < j.nf ? __dadd_rn(f_lo, __dmul_rn(25.0, (double)(k0 + f))) : 0.0;
        j.freq = j.fr[0];
        mine[i] = j;
    }
}

// per detected PRN: the fine frequency with the largest (non-coherent) sum; the first one on ties (B1C :289-296, B2a :318-325).
// One workgroup (256 threads) per PRN: the magnitudes of all (segment, component, frequency) sums in parallel, then one
// thread per frequency adds its segments in order, then the first maximum.
__global__ __launch_bounds__(256) void k_ref_fine_pick(RefParams p, RefPrn *__restrict__ prn, const double2 *__restrict__ jobout, int slices) {
    const int pi = blockIdx.x, tid = threadIdx.x;
    RefPrn &r = prn[pi];
    if (!r.detected || (r.flags & kRefFineRange)) return;  // (workgroup-uniform)
    const bool b1c = p.signal == BDS_SIGNAL_B1C;
    const int nseg = b1c ? 1 : p.fineNoncoh, ncp = b1c ? p.ncomp : 2;
    const size_t job0 = (size_t)r.fine_job0;
    extern __shared__ double s_mag[];  // [seg][comp][kf]
    double *const s_val = s_mag + nseg * ncp * p.nfine;  // [kf]
    const int total = nseg * ncp * p.nfine;
    for (int i = tid; i < total; i += 256) {
        const int kf = i % p.nfine, comp = (i / p.nfine) % ncp, seg = i / (p.nfine * ncp);
        const size_t j = job0 + ((size_t)seg * p.nchunk + kf / kCorrFreqs) * ncp + comp;
        double2 acc = make_double2(0.0, 0.0);
        for (int k = 0; k < slices; ++k) {
            const double2 t = jobout[(j * slices + k) * kCorrFreqs + kf % kCorrFreqs];
            acc.x = __dadd_rn(acc.x, t.x), acc.y = __dadd_rn(acc.y, t.y);
        }
        s_mag[i] = ref_cabs(acc);
    }
    __syncthreads();
    if (tid < p.nfine) {
        const int kf = tid;
        double v;
        if (b1c) {
            v = s_mag[kf];
            if (p.ncomp == 2) v = __ddiv_rn(__dadd_rn(__dmul_rn(v, 11.0), __dmul_rn(s_mag[p.nfine + kf], 29.0)), 40.0);  // :291-292
        } else {
            double sd = 0.0, sp = 0.0;
            for (int seg = 0; seg < nseg; ++seg)
                sd = __dadd_rn(sd, s_mag[(seg * 2 + 0) * p.nfine + kf]), sp = __dadd_rn(sp, s_mag[(seg * 2 + 1) * p.nfine + kf]);
            v = __dadd_rn(sd, sp);  // :321
        }
        s_val[kf] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double best = -1.0;
        int kbest = 0;
        for (int kf = 0; kf < p.nfine; ++kf)
            if (s_val[kf] > best) best = s_val[kf], kbest = kf;
        r.kbest = kbest;
    }
}

}  // namespace bds
