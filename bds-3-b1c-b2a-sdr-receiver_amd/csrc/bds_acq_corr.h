// f64 coherent sums of the refinement (included by bds_acq_kernels.h): the values that DECIDE -- peakSize, the B2a second
// peak, the fine-Doppler ranking -- are direct time-domain correlations in f64 of the few (bin, lag) cells the sieve hands over
// (B2a/acquisition.m:203-209,287-321; B1C/acquisition.m:207-219,276-292).
//
// Round 5: one kernel family k_corr<NC, FM, KIND> instead of k_corr_f64 / k_corr_f64_multi.
//   * NC jobs that differ only in their code slot (the data and pilot component of one candidate / one fine-search segment,
//     adjacent in the job list) are summed in ONE pass: a sample is loaded, mean-corrected and converted once, the carrier
//     phasor of a frequency is advanced once for both components (it was half of a frequency's arithmetic);
//   * FM = 1: the job's single frequency jb.freq; FM = kCorrFreqs: up to six fine-search frequencies jb.fr[] that share the
//     job's samples and code;
//   * the sample loop runs in blocks with an exact f64 phasor at the head of each block (and of the piece behind the circular
//     wrap, where the time index jumps) and constant-angle rotations in between; a thread takes four CONSECUTIVE samples per step
//     with one unaligned dword load per stream (record, data code, pilot code) and the next step's loads are requested before
//     this step's arithmetic: the old loop's byte load -> convert -> accumulate chain was a latency chain of ~50 dependent
//     global loads per thread, and byte loads alone (three vector-memory instructions per sample) cost more issue time than the
//     f64 arithmetic (cfg2: 0.80 ms for a fine search whose arithmetic is under 0.1 ms);
//   * the record type (real / interleaved I/Q int8, f64 after the resampling branch) is a template parameter: no switch and no
//     zero Q arithmetic in the loop of a real record;
//   * the 256 partial sums are added by wave (xor butterflies) and the four wave sums in order by one thread: one barrier per
//     job instead of eight per frequency.  The order is fixed, so a job's sums are the same bits wherever it runs.
#pragma once

namespace bds {

struct CorrJob {
    long start;     // first sample (0-based)
    long len;       // samples to sum
    long code_k0;   // mode 1: offset of this segment inside the long code
    double freq;    // [Hz]
    double mean;    // subtracted from every sample (B1C/acquisition.m:254), else 0
    double mean_q;  // ... from the Q part of a complex sample
    int slot;       // code slot (prn_idx*ncomp + comp)
    int circ;
    int mode;
    int nf;         // FM > 1: frequencies of this job (1 .. kCorrFreqs), fr[0 .. nf); 0 = place-holder (skipped)
    double fr[6];   // ... the fine-search frequencies that share the job's samples and code [Hz]
};
constexpr int kCorrFreqs = 6;

// kQ consecutive samples of the record as loaded -- one 4- / 8-byte load for an int8 record instead of four byte loads (a vector
// memory instruction costs its wave 100-300 cycles of issue time on a busy CU whatever its width: at one byte per lane and step
// the loads, not the f64 arithmetic, set the kernel's time) -- converted where they are used, so that the load stays in flight
// behind the previous group's arithmetic.  The addresses are unaligned (code phase + segment offsets): byte-wise copies that
// the compiler turns into single unaligned dword loads (gfx950 handles them in hardware).
constexpr int kQ = 4;
template <int KIND>
struct CorrQuad;
template <>
struct CorrQuad<kS8> {
    unsigned r;
    __device__ __forceinline__ void load(const void *p, long m) { __builtin_memcpy(&r, reinterpret_cast<const int8_t *>(p) + m, 4); }
    __device__ __forceinline__ void get(int j, double &x, double &xq) const { x = (double)(int)(int8_t)(r >> (8 * j)), xq = 0.0; }
};
template <>
struct CorrQuad<kS8C> {
    unsigned r[2];  // I, Q, I, Q | I, Q, I, Q
    __device__ __forceinline__ void load(const void *p, long m) { __builtin_memcpy(r, reinterpret_cast<const int8_t *>(p) + 2 * m, 8); }
    __device__ __forceinline__ void get(int j, double &x, double &xq) const {
        const unsigned w = r[j >> 1] >> (16 * (j & 1));
        x = (double)(int)(int8_t)(w & 0xff), xq = (double)(int)(int8_t)((w >> 8) & 0xff);
    }
};
template <>
struct CorrQuad<kF64> {
    double r[kQ];
    __device__ __forceinline__ void load(const void *p, long m) {
#pragma unroll
        for (int j = 0; j < kQ; ++j) r[j] = reinterpret_cast<const double *>(p)[m + j];
    }
    __device__ __forceinline__ void get(int j, double &x, double &xq) const { x = r[j], xq = 0.0; }
};
template <>
struct CorrQuad<kF64C> {
    double2 r[kQ];
    __device__ __forceinline__ void load(const void *p, long m) {
#pragma unroll
        for (int j = 0; j < kQ; ++j) r[j] = reinterpret_cast<const double2 *>(p)[m + j];
    }
    __device__ __forceinline__ void get(int j, double &x, double &xq) const { x = r[j].x, xq = r[j].y; }
};

// a wave-uniform f64 value into scalar registers (the step rotations: 24 VGPRs at six frequencies)
__device__ __forceinline__ double corr_uniform(double v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(v)), hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}

__device__ __forceinline__ double corr_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// grid (units, slices): unit u = jobs[u*NC .. u*NC + NC) (same samples, frequency and mode; slots differ), cut into gridDim.y
// contiguous slices whose partial sums the consumer adds in order.
//   out[((u*NC + c) * slices + slice) * FM + f]
// njobs_dev != nullptr: the unit count lives on the device (the refinement chain of bds_acq_refine.h builds its jobs there and
// the host never learns the count before the launch): a fixed grid walks min(*njobs_dev, dev_cap) units.
// Thread t of a step takes the kQ consecutive samples n0 + kQ t .. + kQ - 1; a step of the workgroup covers 256 kQ samples.  The
// phasor of a frequency runs as ONE chain: exact at the head of a block, one-sample rotations through the thread's kQ samples,
// a (256 kQ - (kQ - 1))-sample rotation to the next step; kBlk steps per block = 4 kBlk rotations from an exact value (16 for the
// sums that decide, 64 for the fine search, whose sums only rank frequencies: ~7e-15).
template <int NC, int FM, int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_corr(SampleView sig, long n_circ, const int8_t *__restrict__ codes,
                                                                                    long code_stride, double inv_fs,
                                                                                    const CorrJob *__restrict__ jobs, double2 *__restrict__ out,
                                                                                    const int *__restrict__ njobs_dev, int dev_cap) {
    constexpr bool CPLX = KIND == kS8C || KIND == kF64C;
    constexpr int kBlk = FM > 1 ? 16 : 4;  // steps between exact phasors
    constexpr long kSpan = 256L * kQ;      // samples per step of the workgroup
    const long nunits = njobs_dev ? (long)min(*njobs_dev, dev_cap) : (long)gridDim.x;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double s_part[4][NC * FM * 2];
    __shared__ double s_rot[2][FM][2], s_fr[FM];  // [0]: one-sample rotation, [1]: the rotation to the next step
    // FM > 1: the exact phasor at the head of a block is (phasor of the wave's lane 0 there) x (phasor of the lane's offset):
    // FM lanes of a wave evaluate the first, a table per unit holds the second -- one f64 sincospi stream per wave and block
    // instead of FM per thread (inlined six times they also cost the kernel its registers: 256 + scratch, one wave per SIMD)
    __shared__ double2 s_lanep[FM > 1 ? FM : 1][64];
    __shared__ double2 s_base[4][FM];
    for (long ux = blockIdx.x; ux < nunits; ux += gridDim.x) {
        const CorrJob *const jp = jobs + ux * NC;
        const int nf = FM > 1 ? jp->nf : 1;
        if (nf <= 0) continue;  // a place-holder job (the host path lists none): workgroup-uniform
        const long jb_start = jp->start, jb_len = jp->len, jb_k0 = jp->code_k0;
        const int jb_circ = jp->circ, jb_mode = jp->mode;
        const double jb_mean = jp->mean, jb_mean_q = jp->mean_q;
        const long slice = ((jb_len + gridDim.y - 1) / gridDim.y + 255) & ~255L;
        const long n_lo = (long)blockIdx.y * slice, n_hi = n_lo + slice < jb_len ? n_lo + slice : jb_len;
        // the rotations of each frequency are the same for every thread: 2 FM lanes evaluate them, LDS hands them round
        if (tid < 2 * FM) {
            const int f = tid % FM, which = tid / FM;
            const double fq = FM > 1 ? (f < nf ? jp->fr[f] : 0.0) : jp->freq;
            if (which == 0) s_fr[f] = fq;
            const double dcyc = fq * ((which ? (double)(kSpan - (kQ - 1)) : 1.0) * inv_fs);
            sincospi(2.0 * (dcyc - floor(dcyc)), &s_rot[which][f][1], &s_rot[which][f][0]);
        }
        __syncthreads();
        if constexpr (FM > 1) {
            for (int e = tid; e < 64 * FM; e += 256) {
                const double cyc = s_fr[e >> 6] * ((double)(kQ * (e & 63)) * inv_fs);
                double sn, cs;
                sincospi(2.0 * (cyc - floor(cyc)), &sn, &cs);
                s_lanep[e >> 6][e & 63] = make_double2(cs, sn);
            }
            __syncthreads();
        }
        double sr[NC][FM], si[NC][FM], w1r[FM], w1i[FM], wsr[FM], wsi[FM], cr[FM], ci[FM];
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            w1r[f] = corr_uniform(s_rot[0][f][0]), w1i[f] = corr_uniform(s_rot[0][f][1]);
            wsr[f] = corr_uniform(s_rot[1][f][0]), wsi[f] = corr_uniform(s_rot[1][f][1]);
            cr[f] = 1.0, ci[f] = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) sr[c][f] = si[c][f] = 0.0;
        }
        const double fq1 = corr_uniform(s_fr[0]);
        const int8_t *cp[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int slot = jp[c].slot;
            BDS_DASSERT(slot >= 0 && slot < 2 * BDS_MAX_PRN);
            cp[c] = codes + ((long)slot * 2 + jb_mode) * code_stride + (jb_mode ? jb_k0 : 0);
        }
        // a circular job (the coarse sums: a = start + n wraps at N, the time index is the wrapped one) is two linear pieces
        const long n_wrap = jb_circ ? n_circ - jb_start : jb_len;  // first n behind the wrap
#pragma unroll 1
        for (int piece = 0; piece < 2; ++piece) {
            const long p_lo = piece == 0 ? n_lo : (n_lo > n_wrap ? n_lo : n_wrap);
            const long p_hi = piece == 0 ? (n_hi < n_wrap ? n_hi : n_wrap) : n_hi;
            if (p_lo >= p_hi) continue;
            const long a_off = jb_start - (piece ? n_circ : 0);  // sample index = a_off + n
            const long t_off = jb_circ ? a_off : 0;              // time index = t_off + n (a linear job counts from its first sample)
#pragma unroll 1
            // (a wave runs a block when its FIRST lane has samples there -- wave-uniform: the lanes that evaluate the block's exact
            //  phasors are in it whenever any lane is; a lane without samples runs zero steps)
            for (long nb = p_lo + kQ * tid; nb - kQ * lane < p_hi; nb += kSpan * kBlk) {
                if constexpr (FM > 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    if (lane < FM) {
                        const double cyc = s_fr[lane] * ((double)(t_off + nb - kQ * lane) * inv_fs);
                        double sn, cs;
                        sincospi(2.0 * (cyc - floor(cyc)), &sn, &cs);
                        s_base[wave][lane] = make_double2(cs, sn);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int f = 0; f < FM; ++f) {
                        const double2 b = s_base[wave][f], l = s_lanep[f][lane];
                        cr[f] = b.x * l.x - b.y * l.y;
                        ci[f] = b.x * l.y + b.y * l.x;
                    }
                } else {
                    const double cyc = fq1 * ((double)(t_off + nb) * inv_fs);
                    sincospi(2.0 * (cyc - floor(cyc)), &ci[0], &cr[0]);
                }
                const long rem = nb < p_hi ? (p_hi - nb + kSpan - 1) / kSpan : 0;  // steps this thread has left in the piece
                const int steps = rem < kBlk ? (int)rem : kBlk;
                // the raw bytes of the NEXT step are requested before this step's arithmetic; a request behind the end of the piece
                // re-reads the piece's first samples and counts as zero (no branch in the loop; a group that straddles the end reads
                // up to kQ - 1 samples past it: the next piece / segment, or the padding behind the block and the code tables)
                CorrQuad<KIND> cur, nxt;
                unsigned cv[NC], cvn[NC];
                auto request = [&](int i, CorrQuad<KIND> &xs, unsigned(&cs)[NC]) {
                    const long n = nb + kSpan * i;
                    const long nc_ = n < p_hi ? n : p_lo;
                    BDS_DASSERT(nc_ >= 0 && (jb_mode ? jb_k0 + nc_ : nc_) < code_stride);
                    xs.load(sig.p, a_off + nc_);
#pragma unroll
                    for (int c = 0; c < NC; ++c) __builtin_memcpy(&cs[c], cp[c] + nc_, 4);
                };
                request(0, cur, cv);
#pragma unroll 1
                for (int i = 0; i < steps; ++i) {
                    request(i + 1, nxt, cvn);
                    const long n0 = nb + kSpan * i;
#pragma unroll
                    for (int q = 0; q < kQ; ++q) {
                        const bool live = n0 + q < p_hi;
                        double x, xq;
                        cur.get(q, x, xq);
                        const double xm = live ? x - jb_mean : 0.0;
                        const double xqm = CPLX ? (live ? xq - jb_mean_q : 0.0) : 0.0;
                        double xc[NC], xqc[NC];
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const double cvd = (double)(int)(int8_t)(cv[c] >> (8 * q));
                            xc[c] = xm * cvd, xqc[c] = xqm * cvd;
                        }
#pragma unroll
                        for (int f = 0; f < FM; ++f) {  // (all FM: a frequency beyond nf is 0 Hz, its sums are never written)
                            if (q > 0) {
                                const double nr = cr[f] * w1r[f] - ci[f] * w1i[f];
                                ci[f] = cr[f] * w1i[f] + ci[f] * w1r[f];
                                cr[f] = nr;
                            }
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                if constexpr (CPLX) {
                                    sr[c][f] += xc[c] * cr[f] - xqc[c] * ci[f];
                                    si[c][f] += xc[c] * ci[f] + xqc[c] * cr[f];
                                } else {
                                    sr[c][f] += xc[c] * cr[f];
                                    si[c][f] += xc[c] * ci[f];
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int f = 0; f < FM; ++f) {  // to the first sample of the next step
                        const double nr = cr[f] * wsr[f] - ci[f] * wsi[f];
                        ci[f] = cr[f] * wsi[f] + ci[f] * wsr[f];
                        cr[f] = nr;
                    }
                    cur = nxt;
#pragma unroll
                    for (int c = 0; c < NC; ++c) cv[c] = cvn[c];
                }
            }
        }
        // 256 partial sums per (component, frequency): by wave, then the four wave sums in order
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int f = 0; f < FM; ++f) {
                const double vr = corr_wave_sum(sr[c][f]), vi = corr_wave_sum(si[c][f]);
                if (lane == 0) s_part[wave][(c * FM + f) * 2] = vr, s_part[wave][(c * FM + f) * 2 + 1] = vi;
            }
        __syncthreads();
        if (tid < NC * FM) {
            const int c = tid / FM, f = tid % FM;
            if (f < nf) {
                double vr = s_part[0][tid * 2], vi = s_part[0][tid * 2 + 1];
#pragma unroll
                for (int w = 1; w < 4; ++w) vr += s_part[w][tid * 2], vi += s_part[w][tid * 2 + 1];
                out[(((size_t)ux * NC + c) * gridDim.y + blockIdx.y) * FM + f] = make_double2(vr, vi);
            }
        }
        __syncthreads();  // s_part / s_rot are re-used by the next unit of this workgroup
    }
}

// host-side dispatch on the record type and the number of components summed together
template <int FM>
static void launch_corr(hipStream_t s_, dim3 grid, const SampleView &sig, int nc, long n_circ, const int8_t *codes, long code_stride,
                        double inv_fs, const CorrJob *jobs, double2 *out, const int *njobs_dev, int dev_cap) {
#define BDS_CORR_GO(NC_, K_) \
    hipLaunchKernelGGL((k_corr<NC_, FM, K_>), grid, dim3(256), 0, s_, sig, n_circ, codes, code_stride, inv_fs, jobs, out, njobs_dev, dev_cap)
#define BDS_CORR_KIND(NC_)                        \
    switch (sig.kind) {                           \
        case kS8: BDS_CORR_GO(NC_, kS8); break;   \
        case kS8C: BDS_CORR_GO(NC_, kS8C); break; \
        case kF64: BDS_CORR_GO(NC_, kF64); break; \
        default: BDS_CORR_GO(NC_, kF64C); break;  \
    }
    if (nc == 2) {
        BDS_CORR_KIND(2)
    } else {
        BDS_CORR_KIND(1)
    }
#undef BDS_CORR_KIND
#undef BDS_CORR_GO
}

}  // namespace bds
