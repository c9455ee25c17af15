// Input conditioning of the acquisition's optional resampling branch
// (B2a/acquisition.m:56-124, B1C/acquisition.m:56-123): band-pass fir1(700) + zero-phase
// filtfilt around the IF, band-pass-sampling rate choice, index decimation, IF' = rem(IF, fs').
// Everything runs in f64 like the reference; the conditioned block stays in HBM as f64
// (real, or interleaved complex for a fileType-2 record) and the search kernels read it through
// the same SampleView as the raw int8 block.
#pragma once
#include "bds_debug.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "bds_mi355x.h"

namespace bds {

// ---- sample storage seen by the search kernels ---------------------------------------------
enum SampleKind : int { kS8 = 0, kS8C = 1, kF64 = 2, kF64C = 3 };

struct SampleView {
    const void *p;
    int kind;
    long n = 0;  // samples behind p (debug build: bound of every load; 0 = unknown)
    __device__ __forceinline__ double2 load(long m) const {
        BDS_DASSERT(m >= 0 && (n == 0 || m < n));
        switch (kind) {
            case kS8: return make_double2((double)reinterpret_cast<const int8_t *>(p)[m], 0.0);
            case kS8C: {
                const char2 v = reinterpret_cast<const char2 *>(p)[m];
                return make_double2((double)v.x, (double)v.y);
            }
            case kF64: return make_double2(reinterpret_cast<const double *>(p)[m], 0.0);
            default: return reinterpret_cast<const double2 *>(p)[m];
        }
    }
    __host__ __device__ bool is_complex() const { return kind == kS8C || kind == kF64C; }
};

// ---- what the branch changes in the settings ------------------------------------------------
struct ResamplePlan {
    bool on = false;
    double old_fs = 0, old_if = 0;  // oldFreq, oldIF (acquisition.m:99,116)
    double new_fs = 0, new_if = 0;  // settings.samplingFreq / settings.IF inside acquisition()
    double wp1 = 0, wp2 = 0;        // fir1 band edges (fraction of Nyquist, :66)
};

inline ResamplePlan resample_plan(const bds_settings &s) {
    ResamplePlan r;
    if (!(s.samplingFreq > s.resamplingThreshold && s.resamplingflag == 1)) return r;  // :54-55
    r.on = true;
    const double fs = s.samplingFreq, IF = s.IF;
    const double bw = s.signal == BDS_SIGNAL_B1C ? 9e6 : s.codeFreqBasis * 2 + 0.5e6;  // B1C :62 / B2a :62
    const double w1 = IF - bw / 2, w2 = IF + bw / 2;
    r.wp1 = w1 * 2 / fs - 0.002;  // :66
    r.wp2 = w2 * 2 / fs + 0.002;
    const double fu = IF + bw / 2;  // :77
    double n = std::floor(fu / bw);
    if (n < 1) n = 1;
    const double lower = 2 * fu / n;
    const double fl = IF - bw / 2;
    const double upper = n > 1 ? 2 * fl / (n - 1) : lower;
    r.old_fs = fs;
    r.old_if = IF;
    r.new_fs = std::ceil((lower + upper) / 2);  // :103
    r.new_if = std::fmod(IF, r.new_fs);         // :119 rem()
    return r;
}

// b = fir1(n_taps - 1, [wp1 wp2]): band-pass window design, Hamming window, gain 1 at the
// centre of the pass band (fir1's default 'scale').
inline std::vector<double> fir1_bandpass(int n_taps, double wp1, double wp2) {
    const double pi = 3.14159265358979323846;
    std::vector<double> h((size_t)n_taps);
    const double alpha = 0.5 * (n_taps - 1);
    auto sinc = [&](double x) { return x == 0.0 ? 1.0 : std::sin(pi * x) / (pi * x); };
    for (int i = 0; i < n_taps; ++i) {
        const double m = (double)i - alpha;
        const double ideal = wp2 * sinc(wp2 * m) - wp1 * sinc(wp1 * m);
        const double win = 0.54 - 0.46 * std::cos(2.0 * pi * (double)i / (double)(n_taps - 1));
        h[(size_t)i] = ideal * win;
    }
    const double f0 = 0.5 * (wp1 + wp2);
    double sc = 0;
    for (int i = 0; i < n_taps; ++i) sc += h[(size_t)i] * std::cos(pi * ((double)i - alpha) * f0);
    for (double &v : h) v /= std::fabs(sc);
    return h;
}

// ---- kernels ----------------------------------------------------------------------------------
// xt = [2*x(1)-x(nfact+1:-1:2); x; 2*x(end)-x(end-1:-1:end-nfact)]  (filtfilt's edge extension);
// NCH = 1 real, 2 interleaved complex; the int8 record is widened to f64 here.
template <int NCH>
__global__ __launch_bounds__(256) void k_ff_extend(const int8_t *__restrict__ x, long n, int nfact,
                                                   double *__restrict__ e) {
    const long tot = n + 2L * nfact;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double v;
            if (i < nfact)
                v = 2.0 * (double)x[0 * NCH + c] - (double)x[(long)(nfact - i) * NCH + c];
            else if (i < nfact + n)
                v = (double)x[(i - nfact) * NCH + c];
            else
                v = 2.0 * (double)x[(n - 1) * NCH + c] - (double)x[(n - 2 - (i - nfact - n)) * NCH + c];
            e[i * NCH + c] = v;
        }
    }
}

// One direction of filtfilt: y = filter(b, 1, u, zi*u(1)) with the steady-state initial condition,
// i.e. u(m) = u(1) for m < 1.  reverse != 0: u is `in` read back to front and y is written back to
// front (the second, time-reversed pass).  Taps are summed from the oldest sample to the newest,
// the order a transposed direct-form filter produces.
template <int NCH>
__global__ __launch_bounds__(256) void k_ff_fir(const double *__restrict__ in, long len,
                                                const double *__restrict__ b, int n_taps, int reverse,
                                                double *__restrict__ out) {
    extern __shared__ double s_b[];
    for (int i = threadIdx.x; i < n_taps; i += blockDim.x) s_b[i] = b[i];
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long)gridDim.x * blockDim.x) {
        double acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = 0.0;
        for (int k = n_taps - 1; k >= 0; --k) {
            long j = i - k;
            if (j < 0) j = 0;
            const long src = reverse ? len - 1 - j : j;
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = fma(s_b[k], in[src * NCH + c], acc[c]);
        }
        const long dst = reverse ? len - 1 - i : i;
#pragma unroll
        for (int c = 0; c < NCH; ++c) out[dst * NCH + c] = acc[c];
    }
}

// longSignal = longSignal(index), index = ceil((0:signalLen-1)/fs' * oldFreq), index(1) = 1
// (:107-112); z is the filtered, still extended signal (offset nfact).
template <int NCH>
__global__ __launch_bounds__(256) void k_ff_decimate(const double *__restrict__ z, int nfact, long sig_len,
                                                     double new_fs, double old_fs, double *__restrict__ out) {
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < sig_len; k += (long)gridDim.x * blockDim.x) {
        long idx = (long)ceil(((double)k / new_fs) * old_fs);
        if (k == 0) idx = 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) out[k * NCH + c] = z[(idx - 1 + nfact) * NCH + c];
    }
}

}  // namespace bds
