// Device kernels of the acquisition search (included by bds_acq.hip).
//
// Data layout in HBM (all fp32 complex unless noted):
//   sig      int8  [n_samples]                 IF block as read from the file
//   Xs       c64   [D][L]      spectrum of the carrier-wiped, periodically extended block
//                              per Doppler bin, [k1][k2] order
//   Cs       c64   [P][ncomp][L]  conj(code spectrum)/L per PRN and component, same order
//   Bw       c64   [G][ncomp][L]  work: forward column-pass output / inverse row-pass output
//   recs     {f32 value, i32 lag} [P][D][tiles]  per-workgroup maxima of the column pass
//
// Reference lines each kernel replaces are cited at the kernel.
#pragma once

#include "bds_fft.h"
#include "bds_resample.h"

namespace bds {

// ---- loaders ---------------------------------------------------------------------

// Carrier wipe-off of the int8 block for Doppler bin `batch`, evaluated at extended
// index n (B2a/acquisition.m:146,194-201; B1C/acquisition.m:144,198-205):
//   y[n] = x[n mod N] * exp(+j 2 pi f_b (n mod N) / fs),  n < N + X - 1, else 0.
// The phase is reduced to [0,1) cycles in f64 (the reference's f*(n*2*pi*ts) reaches
// 1.8e6 rad: an fp32 ramp, as B1C/GPU_acquisition.m:150,206 uses, is off by ~1e-2 rad).
struct SignalLoader {
    SampleView sig;   // int8 record as read from the file, or the f64 block the resampling branch leaves
    long n_circ;      // N
    long n_ext;       // N + X - 1
    double f0, fstep; // bin frequency = f0 + fstep*batch  [Hz]
    double inv_fs;    // 1/fs
    int bin0;         // first bin of this launch
    __device__ __forceinline__ float2 operator()(int batch, long n) const {
        if (n >= n_ext) return make_float2(0.f, 0.f);
        const long m = n < n_circ ? n : n - n_circ;
        // x = I + 1i*Q for a complex record (fileType 2, postProcessing.m:92-96)
        const double2 xv = sig.load(m);
        const float x = (float)xv.x, xq = (float)xv.y;
        const double f = f0 + fstep * (double)(bin0 + batch);
        const double cyc = f * ((double)m * inv_fs);
        const double fr = cyc - floor(cyc);
        const float hi = (float)fr;
        const float lo = (float)(fr - (double)hi);
        float s, c;
        sincospif(2.0f * hi, &s, &c);
        const float d = 6.28318530717958647692f * lo;  // first-order correction for the fp32 rounding of fr
        const float cc = c - d * s, ss = s + d * c;
        return make_float2(x * cc - xq * ss, x * ss + xq * cc);
    }
    // The same value from a carrier phasor the caller maintains: exact() gives exp(+j 2 pi f_b m / fs)
    // in f64, step() the constant rotation between two samples dn apart, mix() applies a phasor to
    // extended index n (used by the specialised column pass, whose threads walk n in equal steps).
    static constexpr bool kHasCarrier = true;
    __device__ __forceinline__ double freq(int batch) const { return f0 + fstep * (double)(bin0 + batch); }
    __device__ __forceinline__ void exact(int batch, long n, double *c, double *s) const {
        const long m = n < n_circ ? n : n - n_circ;
        const double cyc = freq(batch) * ((double)m * inv_fs);
        sincospi(2.0 * (cyc - floor(cyc)), s, c);
    }
    __device__ __forceinline__ void step(int batch, long dn, double *c, double *s) const {
        const double cyc = freq(batch) * ((double)dn * inv_fs);
        sincospi(2.0 * (cyc - floor(cyc)), s, c);
    }
    // fp32 phasors from the f64-reduced phase (round 4, forward column pass): the turn count is reduced in f64 as above, the
    // sine / cosine taken in fp32 with the first-order correction for the rounding of the fraction -- ~1e-7, a third of the
    // instructions of an f64 sincospi, which made that kernel vector-bound (five per thread for twelve elements)
    static __device__ __forceinline__ void phasor32(double cyc, float *c, float *s) {
        const double fr = cyc - floor(cyc);
        const float hi = (float)fr;
        const float lo = (float)(fr - (double)hi);
        float s0, c0;
        sincospif(2.0f * hi, &s0, &c0);
        const float d = 6.28318530717958647692f * lo;
        *c = c0 - d * s0;
        *s = s0 + d * c0;
    }
    __device__ __forceinline__ void exact32(int batch, long n, float *c, float *s) const {
        const long m = n < n_circ ? n : n - n_circ;
        phasor32(freq(batch) * ((double)m * inv_fs), c, s);
    }
    __device__ __forceinline__ void step32(int batch, long dn, float *c, float *s) const {
        phasor32(freq(batch) * ((double)dn * inv_fs), c, s);
    }
    __device__ __forceinline__ float2 mix(long n, double c, double s) const {
        if (n >= n_ext) return make_float2(0.f, 0.f);
        const double2 xv = sig.load(n < n_circ ? n : n - n_circ);
        const float x = (float)xv.x, xq = (float)xv.y, cc = (float)c, ss = (float)s;
        return make_float2(x * cc - xq * ss, x * ss + xq * cc);
    }
};

// Code table value at sample n (0-based), n < X, zero beyond
// (makeB2aDataTable.m:59-67, makeDataTable.m:59-68 + the zero padding of
//  B2a/acquisition.m:179-180, B1C/acquisition.m:176-177,185-186).
struct CodeTable {
    const int8_t *prim;  // [P][ncomp][10230] primary codes (+-1)
    double ts, tc;       // 1/fs ; chip (B2a) or half-chip (B1C) period, as the reference computes them
    long spc;            // samplesPerCode
    long xlen;           // number of non-zero table samples used (X)
    int code_len;        // 10230
    int boc;             // 1: B1C BOC(1,1) half-chips [-c,+c]; 0: B2a
    __device__ __forceinline__ float value(int slot, long n) const {
        if (n >= xlen) return 0.f;
        long idx = (long)ceil((ts * (double)(n + 1)) / tc);  // 1-based chip / half-chip index
        if (n == spc - 1) idx = boc ? 2L * code_len : code_len;
        if (boc && n == 0) idx = 1;
        BDS_DASSERT(slot >= 0 && slot < 2 * BDS_MAX_PRN && idx >= 1 && idx <= (boc ? 2L : 1L) * code_len);
        const int8_t *c = prim + (long)slot * code_len;
        if (boc) {
            const long h = idx - 1;
            const float v = (float)c[h >> 1];
            return (h & 1) ? v : -v;
        }
        return (float)c[idx - 1];
    }
};

struct CodeLoader {
    CodeTable tab;
    int slot0;
    __device__ __forceinline__ float2 operator()(int batch, long n) const {
        return make_float2(tab.value(slot0 + batch, n), 0.f);
    }
};

// ---- forward column pass (first half of fft(), B2a/acquisition.m:183,201) ------------
// grid (tiles, batch); dynamic LDS = T*Spad*8 bytes.
template <class Loader>
__global__ __launch_bounds__(1024) void k_cols_fwd(Plan1D p, TwiddleL twl, int L2, int logT, int Spad,
                                                   Loader ld, float2 *__restrict__ out, long out_stride) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int T = 1 << logT;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int batch = blockIdx.y;
    const int c0 = tile << logT;
    const int S = p.S;
    for (int e = tid; e < (S << logT); e += nthr) {
        const int r = e >> logT, j = e & (T - 1);
        const int col = c0 + j;
        float2 v = make_float2(0.f, 0.f);
        if (col < L2) v = ld(batch, (long)r * L2 + col);
        lds[j * Spad + lds_phys(r)] = v;
    }
    __syncthreads();
    fft_lds<-1>(lds, p, Spad, T, tid, nthr);
    float2 *o = out + (long)batch * out_stride;
    for (int e = tid; e < (S << logT); e += nthr) {
        const int k1 = e >> logT, j = e & (T - 1);
        const int col = c0 + j;
        if (col < L2) {
            const float2 w = twl.get<-1>((uint32_t)k1 * (uint32_t)col);
            o[(long)k1 * L2 + col] = cmul(lds[j * Spad + lds_phys(k1)], w);
        }
    }
}

// ---- forward row pass ------------------------------------------------------------------
// grid (L1 rows, batch); LDS = L2*8.  conj_scale != 0: store conj(X)*scale (code spectra,
// conj(fft(code)) of B2a/acquisition.m:183-184 with ifft's 1/L folded in).
__global__ __launch_bounds__(1024) void k_rows_fwd(Plan1D p, const float2 *__restrict__ in, long in_stride,
                                                   float2 *__restrict__ out, long out_stride,
                                                   int conj_flag, float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int row = blockIdx.x, batch = blockIdx.y;
    const int S = p.S;
    const float2 *src = in + (long)batch * in_stride + (long)row * S;
    for (int e = tid; e < S; e += nthr) lds[lds_phys(e)] = src[e];
    __syncthreads();
    fft_lds<-1>(lds, p, S, 1, tid, nthr);
    float2 *dst = out + (long)batch * out_stride + (long)row * S;
    for (int e = tid; e < S; e += nthr) {
        float2 v = lds[lds_phys(e)];
        v.x *= scale;
        v.y *= conj_flag ? -scale : scale;
        dst[e] = v;
    }
}

// ---- inverse row pass: spectrum product + first half of ifft() ----------------------------
// (IQfreqDom .* conj(codeFreqDom), B2a/acquisition.m:204-205, B1C/acquisition.m:209,216)
// grid (L1 rows, cells); cell g -> bin = bin0 + g (same PRN for the whole launch).
// Writes Bw[g][comp][k1][n2] = twiddle * ifft_row.
template <int NCOMP>
__global__ __launch_bounds__(1024) void k_rows_inv(Plan1D p, TwiddleL twl, const float2 *__restrict__ Xs,
                                                   long L, int bin0, const float2 *__restrict__ Cs /*[ncomp][L] of this PRN*/,
                                                   float2 *__restrict__ Bw) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int k1 = blockIdx.x, g = blockIdx.y;
    const int S = p.S;
    const float2 *xr = Xs + (long)(bin0 + g) * L + (long)k1 * S;
    constexpr int MAXE = kPointsPerThread;
    // inter-pass twiddle W_L^(-k1 e), e = tid + i*nthr, as base(tid) * step(i): the per-element
    // table gather (64 cache lines per instruction) is replaced by one gather per thread and a
    // broadcast LDS read per element.
    __shared__ float2 s_step[MAXE];
    if (tid < MAXE) {
        const long m = (long)k1 * ((long)tid * nthr);
        s_step[tid] = m < L ? twl.get<+1>((uint32_t)m) : make_float2(1.f, 0.f);
    }
    const float2 wbase = tid < S ? twl.get<+1>((uint32_t)k1 * (uint32_t)tid) : make_float2(1.f, 0.f);
    float2 xv[MAXE];
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = tid + i * nthr;
        if (e < S) xv[i] = xr[e];
    }
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
        const float2 *cr = Cs + (long)comp * L + (long)k1 * S;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int e = tid + i * nthr;
            if (e < S) lds[lds_phys(e)] = cmul(xv[i], cr[e]);
        }
        __syncthreads();
        fft_lds<+1>(lds, p, S, 1, tid, nthr);
        float2 *dst = Bw + ((long)g * NCOMP + comp) * L + (long)k1 * S;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int e = tid + i * nthr;
            if (e < S) dst[e] = cmul(lds[lds_phys(e)], cmul(wbase, s_step[i]));
        }
        __syncthreads();
    }
}

struct Rec {
    float v;
    int lag;  // 0-based lag, -1 = none
};

__device__ __forceinline__ void rec_better(float &v, int &lag, float v2, int lag2) {
    // larger value wins; equal values: the smaller lag (MATLAB max returns the first index)
    if (v2 > v || (v2 == v && lag2 >= 0 && (lag < 0 || lag2 < lag))) {
        v = v2;
        lag = lag2;
    }
}

// ---- inverse column pass + |.| combine + maximum -------------------------------------------
// (abs(ifft(.)) [+ abs(ifft(.))] and the max over code phases, B2a/acquisition.m:208-209,
//  218-221; B1C/acquisition.m:212,218-219,229-232.)  The D x N results matrix is never stored.
// grid (tiles, cells).  Lags outside [lo1,hi1] U [lo2,hi2] (0-based, inclusive) are ignored:
// the full search passes [0, N-1] and an empty second range; the B2a second-peak pass the two
// ranges of B2a/acquisition.m:224-246.
template <int NCOMP>
__global__ __launch_bounds__(1024) void k_cols_inv_max(Plan1D p, int L2, int logT, int Spad,
                                                       const float2 *__restrict__ Bw, long L, float w0,
                                                       float w1, int lo1, int hi1, int lo2, int hi2,
                                                       Rec *__restrict__ recs, int rec_stride) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    __shared__ float s_v[16];
    __shared__ int s_l[16];
    const int T = 1 << logT;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int g = blockIdx.y;
    const int c0 = tile << logT;
    const int S = p.S;
    constexpr int MAXE = kPointsPerThread;
    float mag[MAXE];
#pragma unroll
    for (int comp = 0; comp < NCOMP; ++comp) {
        const float2 *src = Bw + ((long)g * NCOMP + comp) * L;
        for (int e = tid; e < (S << logT); e += nthr) {
            const int r = e >> logT, j = e & (T - 1);
            const int col = c0 + j;
            float2 v = make_float2(0.f, 0.f);
            if (col < L2) v = src[(long)r * L2 + col];
            lds[j * Spad + lds_phys(r)] = v;
        }
        __syncthreads();
        fft_lds<+1>(lds, p, Spad, T, tid, nthr);
        const float w = comp == 0 ? w0 : w1;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int e = tid + i * nthr;
            if (e < (S << logT)) {
                const int n1 = e >> logT, j = e & (T - 1);
                const float2 y = lds[j * Spad + lds_phys(n1)];
                const float a = w * sqrtf(y.x * y.x + y.y * y.y);
                mag[i] = comp == 0 ? a : mag[i] + a;
            }
        }
        __syncthreads();
    }
    float bv = -1.f;
    int bl = -1;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = tid + i * nthr;
        if (e < (S << logT)) {
            const int n1 = e >> logT, j = e & (T - 1);
            const int col = c0 + j;
            const long lag = (long)n1 * L2 + col;
            const bool in = col < L2 && ((lag >= lo1 && lag <= hi1) || (lag >= lo2 && lag <= hi2));
            if (in) rec_better(bv, bl, mag[i], (int)lag);
        }
    }
    // wave reduce (64 lanes), then across waves through LDS
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(bv, off, 64);
        const int ol = __shfl_down(bl, off, 64);
        rec_better(bv, bl, ov, ol);
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) {
        s_v[wave] = bv;
        s_l[wave] = bl;
    }
    __syncthreads();
    if (tid == 0) {
        const int nw = (nthr + 63) >> 6;
        for (int w2 = 1; w2 < nw; ++w2) rec_better(bv, bl, s_v[w2], s_l[w2]);
        Rec r;
        r.v = bv;
        r.lag = bl;
        recs[(long)g * rec_stride + tile] = r;
    }
}

// per cell: reduce the tile records to the row maximum (max(results,[],2) and its lag)
__global__ void k_reduce_rows(const Rec *__restrict__ recs, int rec_stride, int ntiles,
                              float *__restrict__ row_max, int *__restrict__ row_arg) {
    const int g = blockIdx.x;
    float bv = -1.f;
    int bl = -1;
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        const Rec r = recs[(long)g * rec_stride + t];
        rec_better(bv, bl, r.v, r.lag);
    }
    __shared__ float s_v[256];
    __shared__ int s_l[256];
    s_v[threadIdx.x] = bv;
    s_l[threadIdx.x] = bl;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            float v = s_v[threadIdx.x];
            int l = s_l[threadIdx.x];
            rec_better(v, l, s_v[threadIdx.x + s], s_l[threadIdx.x + s]);
            s_v[threadIdx.x] = v;
            s_l[threadIdx.x] = l;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        row_max[g] = s_v[0];
        row_arg[g] = s_l[0];
    }
}

// ---- f64 coherent sums: candidate refinement and the fine-Doppler searches --------------------
// One workgroup per job: out[job] = sum_{n<len} (x[a(n)] - mean) * code(n) * exp(+j 2 pi f t(n)/fs)
//   circ != 0 : a(n) = (start + n) mod n_circ, t(n) = a(n)        (coarse cell, acquisition.m:194-209)
//   circ == 0 : a(n) = start + n,               t(n) = n          (fine search: phase restarts at the
//                                                                  segment, B2a/acquisition.m:287-303,
//                                                                  B1C/acquisition.m:276-287)
//   code mode 0: sampling table (CodeTable::value, n < spc)
//   code mode 1: long code  prim[ floor(ts*(k)/tc) mod code_len ], k = code_k0 + n + 1
//                (B2a/acquisition.m:279-284; always the plain primary code)
// The sampled code of a (slot, mode) is materialised once as int8 (k_make_code, cached in the
// context) so the f64 index arithmetic -- a multiply, a divide and a ceil/floor per sample,
// rounded exactly like the reference -- is not redone for every frequency bin and candidate.
__global__ __launch_bounds__(256) void k_make_code(CodeTable tab, int slot, int mode, long len,
                                                   int8_t *__restrict__ out) {
    for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < len; n += (long)gridDim.x * blockDim.x) {
        float cv;
        if (mode == 0) {
            cv = tab.value(slot, n);
        } else {
            const long ci = (long)floor((tab.ts * (double)(n + 1)) / tab.tc);
            BDS_DASSERT(ci >= 0 && slot >= 0 && slot < 2 * BDS_MAX_PRN);
            cv = (float)tab.prim[(long)slot * tab.code_len + (ci % tab.code_len)];
        }
        out[n] = (int8_t)cv;
    }
}

}  // namespace bds

#include "bds_acq_corr.h"  // CorrJob, k_corr<NC, FM, KIND>: the f64 coherent sums
