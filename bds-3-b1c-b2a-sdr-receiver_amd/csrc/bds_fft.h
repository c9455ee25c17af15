// Mixed-radix Stockham FFT engine held in LDS (gfx950), used by the two-pass
// (rows x columns) transforms of the acquisition search.
//
// A length-L transform (L = L1*L2, 5-smooth) is done in two kernels:
//   column pass : T adjacent columns per workgroup, L1-point transforms, LDS resident
//   row pass    : one contiguous row per workgroup, L2-point transform, LDS resident
// The forward direction runs columns-then-rows and leaves the spectrum in [k1][k2]
// order (element X[k1 + L1*k2] at offset k1*L2 + k2); the inverse runs
// rows-then-columns from that order straight back to natural order, so no
// transpose pass exists anywhere (tools/proto_fft.py checks the index maps).
//
// Every stage is a Stockham autosort radix-R step over T independent transforms
// of length S laid out as buf[j*Spad + i]:  all threads pull their butterfly
// inputs into registers, barrier, twiddle + butterfly, scatter to the autosort
// position, barrier.  One LDS buffer, no ping-pong copy.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace bds {

constexpr int kMaxStages = 12;
constexpr int kPointsPerThread = 16;  // register budget per stage: floor(16/R)*R points

struct FastDiv {
    uint32_t d, m;  // m = ceil(2^32/d), or 0 when d == 1
    __host__ __device__ FastDiv() : d(1), m(0) {}
    __host__ explicit FastDiv(uint32_t dd) : d(dd), m(dd == 1 ? 0u : (uint32_t)((0x100000000ull + dd - 1) / dd)) {}
    // exact for x*d < 2^32
    __device__ __forceinline__ uint32_t div(uint32_t x) const { return m ? __umulhi(x, m) : x; }
};

struct Plan1D {
    int S;       // transform length
    int nstage;  // number of radix stages
    int radix[kMaxStages];
    FastDiv nb[kMaxStages];  // S / radix
    FastDiv ns[kMaxStages];  // product of earlier radices
    int tws[kMaxStages];     // S / (Ns*R): stride into the W_S table
    const float2 *tw;        // device: W_S^i = exp(-2 pi i / S), i < S
};

// W_L^m, m < L, as hi[m >> kTwLoBits] * lo[m & mask]  (both tables rounded from f64)
constexpr int kTwLoBits = 10;
struct TwiddleL {
    const float2 *hi;
    const float2 *lo;
    template <int DIR>
    __device__ __forceinline__ float2 get(uint32_t m) const {
        const float2 a = hi[m >> kTwLoBits];
        const float2 b = lo[m & ((1u << kTwLoBits) - 1)];
        float2 w = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
        if (DIR > 0) w.y = -w.y;  // table is exp(-j..): inverse wants the conjugate
        return w;
    }
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -j*DIR ... i.e. exp(-j pi/2) for the forward transform, +j for the inverse
template <int DIR>
__device__ __forceinline__ float2 rot90(float2 a) {
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

template <int R, int DIR>
struct Butterfly;

template <int DIR>
struct Butterfly<2, DIR> {
    __device__ __forceinline__ static void run(float2 *v) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <int DIR>
struct Butterfly<3, DIR> {
    __device__ __forceinline__ static void run(float2 *v) {
        // X1,2 = a - (b+c)/2 -+ j*DIR... with s = sin(2pi/3)
        const float s = 0.86602540378443864676f;
        const float2 a = v[0], t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
        const float2 m = make_float2(a.x - 0.5f * t.x, a.y - 0.5f * t.y);
        // forward: X1 = m - j s d ; inverse: X1 = m + j s d
        const float2 jd = DIR < 0 ? make_float2(s * d.y, -s * d.x) : make_float2(-s * d.y, s * d.x);
        v[0] = cadd(a, t);
        v[1] = cadd(m, jd);
        v[2] = csub(m, jd);
    }
};

template <int DIR>
struct Butterfly<4, DIR> {
    __device__ __forceinline__ static void run(float2 *v) {
        const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
        const float2 c = cadd(v[1], v[3]), d = rot90<DIR>(csub(v[1], v[3]));
        v[0] = cadd(a, c);
        v[1] = cadd(b, d);
        v[2] = csub(a, c);
        v[3] = csub(b, d);
    }
};

template <int DIR>
struct Butterfly<5, DIR> {
    __device__ __forceinline__ static void run(float2 *v) {
        const float c1 = 0.30901699437494742410f;   // cos(2pi/5)
        const float c2 = -0.80901699437494742410f;  // cos(4pi/5)
        const float s1 = 0.95105651629515357212f;   // sin(2pi/5)
        const float s2 = 0.58778525229247312917f;   // sin(4pi/5)
        const float2 a = v[0];
        const float2 p1 = cadd(v[1], v[4]), m1 = csub(v[1], v[4]);
        const float2 p2 = cadd(v[2], v[3]), m2 = csub(v[2], v[3]);
        const float2 r1 = make_float2(a.x + c1 * p1.x + c2 * p2.x, a.y + c1 * p1.y + c2 * p2.y);
        const float2 r2 = make_float2(a.x + c2 * p1.x + c1 * p2.x, a.y + c2 * p1.y + c1 * p2.y);
        // q1 = s1 m1 + s2 m2 ; q2 = s2 m1 - s1 m2 ; forward: X1 = r1 - j q1, X2 = r2 - j q2
        const float2 q1 = make_float2(s1 * m1.x + s2 * m2.x, s1 * m1.y + s2 * m2.y);
        const float2 q2 = make_float2(s2 * m1.x - s1 * m2.x, s2 * m1.y - s1 * m2.y);
        const float2 jq1 = DIR < 0 ? make_float2(q1.y, -q1.x) : make_float2(-q1.y, q1.x);
        const float2 jq2 = DIR < 0 ? make_float2(q2.y, -q2.x) : make_float2(-q2.y, q2.x);
        v[0] = make_float2(a.x + p1.x + p2.x, a.y + p1.y + p2.y);
        v[1] = cadd(r1, jq1);
        v[4] = csub(r1, jq1);
        v[2] = cadd(r2, jq2);
        v[3] = csub(r2, jq2);
    }
};

// One Stockham stage of radix R over T transforms of length p.S in buf[j*Spad + i].
// Requires (S/R)*T <= floor(16/R) * blockDim.x  (checked on the host).
template <int R, int DIR>
__device__ __forceinline__ void fft_stage(float2 *__restrict__ buf, const Plan1D &p, int st, int Spad,
                                          int T, int tid, int nthr) {
    constexpr int MB = kPointsPerThread / R;
    const FastDiv nbd = p.nb[st];
    const FastDiv nsd = p.ns[st];
    const int nb = (int)nbd.d;
    const int Ns = (int)nsd.d;
    const int total = nb * T;
    const int tws = p.tws[st];
    float2 v[MB][R];
    int jbase[MB], bbv[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int b = tid + i * nthr;
        if (b < total) {
            const int j = (int)nbd.div((uint32_t)b);
            const int bb = b - j * nb;
            jbase[i] = j * Spad;
            bbv[i] = bb;
            const float2 *src = buf + jbase[i] + bb;
#pragma unroll
            for (int q = 0; q < R; ++q) v[i][q] = src[q * nb];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int b = tid + i * nthr;
        if (b < total) {
            const int bb = bbv[i];
            const int hi = (int)nsd.div((uint32_t)bb);
            const int k = bb - hi * Ns;
            if (Ns > 1) {
                const int kt = k * tws;
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    float2 w = p.tw[q * kt];
                    if (DIR > 0) w.y = -w.y;
                    v[i][q] = cmul(v[i][q], w);
                }
            }
            Butterfly<R, DIR>::run(v[i]);
            float2 *dst = buf + jbase[i] + hi * Ns * R + k;
#pragma unroll
            for (int q = 0; q < R; ++q) dst[q * Ns] = v[i][q];
        }
    }
    __syncthreads();
}

// Full in-LDS transform: T transforms of length p.S, natural order in and out.
// The caller has filled buf and passed a barrier.
template <int DIR>
__device__ __forceinline__ void fft_lds(float2 *__restrict__ buf, const Plan1D &p, int Spad, int T,
                                        int tid, int nthr) {
    for (int st = 0; st < p.nstage; ++st) {
        switch (p.radix[st]) {
            case 2: fft_stage<2, DIR>(buf, p, st, Spad, T, tid, nthr); break;
            case 3: fft_stage<3, DIR>(buf, p, st, Spad, T, tid, nthr); break;
            case 4: fft_stage<4, DIR>(buf, p, st, Spad, T, tid, nthr); break;
            default: fft_stage<5, DIR>(buf, p, st, Spad, T, tid, nthr); break;
        }
    }
}

// XCD-contiguous remap: hardware places workgroup b on XCD b % 8; give each XCD a
// contiguous run of tiles so neighbouring tiles (which share 128-B lines) meet in
// the same L2.  Bijective for any n.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u, xcd = b & 7u, i = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

}  // namespace bds
