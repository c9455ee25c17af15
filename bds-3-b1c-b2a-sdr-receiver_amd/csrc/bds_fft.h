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

// LDS transforms are stored with one pad element after every 16: physical index of
// logical element i.  A radix-R autosort stage writes with a lane stride of R elements
// (a multiple of the 128-B bank row for R = 16); the pad turns that into an odd-ish
// stride so the 64 lanes of a ds_write_b64 spread over all banks.
__host__ __device__ __forceinline__ constexpr int lds_phys(int i) { return i + (i >> 4); }
// elements an LDS transform of logical length S occupies
__host__ __device__ __forceinline__ constexpr int lds_span(int S) { return lds_phys(S - 1) + 1; }

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

// ---- complex arithmetic on two storage/compute types ------------------------------------------
//   float2 : fp32 complex
//   h2     : fp16 complex, one packed VGPR; +,-,* and fma map to v_pk_*_f16 (two lanes of a
//            complex value per instruction) -- used by the reduced-precision sieve only
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
// multiply by -j*DIR ... i.e. exp(-j pi/2) for the forward transform, +j for the inverse
template <int DIR>
__device__ __forceinline__ float2 rot90(float2 a) {
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}
// multiply by the forward constant (cr - j ci) (DIR < 0) or its conjugate (DIR > 0)
template <int DIR>
__device__ __forceinline__ float2 mulc(float2 a, float cr, float ci) {
    return DIR < 0 ? make_float2(a.x * cr + a.y * ci, a.y * cr - a.x * ci)
                   : make_float2(a.x * cr - a.y * ci, a.y * cr + a.x * ci);
}

__device__ __forceinline__ h2 hswap(h2 a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ h2 cmul(h2 a, h2 b) {
    const h2 bx = {b.x, b.x}, by = {-b.y, b.y};
    return __builtin_elementwise_fma(hswap(a), by, a * bx);
}
__device__ __forceinline__ h2 cadd(h2 a, h2 b) { return a + b; }
__device__ __forceinline__ h2 csub(h2 a, h2 b) { return a - b; }
__device__ __forceinline__ h2 cscale(h2 a, float s) {
    const h2 ss = {(_Float16)s, (_Float16)s};
    return a * ss;
}
template <int DIR>
__device__ __forceinline__ h2 rot90(h2 a) {
    const h2 sg = DIR < 0 ? h2{(_Float16)1, (_Float16)-1} : h2{(_Float16)-1, (_Float16)1};
    return hswap(a) * sg;
}
template <int DIR>
__device__ __forceinline__ h2 mulc(h2 a, float cr, float ci) {
    const h2 c1 = {(_Float16)cr, (_Float16)cr};
    const h2 c2 = DIR < 0 ? h2{(_Float16)ci, (_Float16)-ci} : h2{(_Float16)-ci, (_Float16)ci};
    return __builtin_elementwise_fma(hswap(a), c2, a * c1);
}

template <int R, int DIR>
struct Butterfly;

template <int DIR>
struct Butterfly<2, DIR> {
    template <class C>
    __device__ __forceinline__ static void run(C *v) {
        const C a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <int DIR>
struct Butterfly<3, DIR> {
    template <class C>
    __device__ __forceinline__ static void run(C *v) {
        // X1,2 = a - (b+c)/2 -+ j s (b-c), s = sin(2pi/3)
        const float s = 0.86602540378443864676f;
        const C a = v[0], t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
        const C m = csub(a, cscale(t, 0.5f));
        const C jd = rot90<DIR>(cscale(d, s));  // forward: -j s d
        v[0] = cadd(a, t);
        v[1] = cadd(m, jd);
        v[2] = csub(m, jd);
    }
};

template <int DIR>
struct Butterfly<4, DIR> {
    template <class C>
    __device__ __forceinline__ static void run(C *v) {
        const C a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
        const C c = cadd(v[1], v[3]), d = rot90<DIR>(csub(v[1], v[3]));
        v[0] = cadd(a, c);
        v[1] = cadd(b, d);
        v[2] = csub(a, c);
        v[3] = csub(b, d);
    }
};

template <int DIR>
struct Butterfly<5, DIR> {
    template <class C>
    __device__ __forceinline__ static void run(C *v) {
        const float c1 = 0.30901699437494742410f;   // cos(2pi/5)
        const float c2 = -0.80901699437494742410f;  // cos(4pi/5)
        const float s1 = 0.95105651629515357212f;   // sin(2pi/5)
        const float s2 = 0.58778525229247312917f;   // sin(4pi/5)
        const C a = v[0];
        const C p1 = cadd(v[1], v[4]), m1 = csub(v[1], v[4]);
        const C p2 = cadd(v[2], v[3]), m2 = csub(v[2], v[3]);
        const C r1 = cadd(a, cadd(cscale(p1, c1), cscale(p2, c2)));
        const C r2 = cadd(a, cadd(cscale(p1, c2), cscale(p2, c1)));
        // forward: X1 = r1 - j q1, X2 = r2 - j q2 ; q1 = s1 m1 + s2 m2 ; q2 = s2 m1 - s1 m2
        const C jq1 = rot90<DIR>(cadd(cscale(m1, s1), cscale(m2, s2)));
        const C jq2 = rot90<DIR>(csub(cscale(m1, s2), cscale(m2, s1)));
        v[0] = cadd(a, cadd(p1, p2));
        v[1] = cadd(r1, jq1);
        v[4] = csub(r1, jq1);
        v[2] = cadd(r2, jq2);
        v[3] = csub(r2, jq2);
    }
};

template <int DIR>
struct Butterfly<8, DIR> {
    // n = 2 n1 + n2, k = k1 + 4 k2:  X[k1 + 4 k2] = sum_n2 W2^(n2 k2) W8^(n2 k1) sum_n1 x[2 n1 + n2] W4^(n1 k1)
    template <class C>
    __device__ __forceinline__ static void run(C *v) {
        const float h = 0.70710678118654752440f;
        C a0[4] = {v[0], v[2], v[4], v[6]};
        C a1[4] = {v[1], v[3], v[5], v[7]};
        Butterfly<4, DIR>::run(a0);
        Butterfly<4, DIR>::run(a1);
        a1[1] = mulc<DIR>(a1[1], h, h);    // W8^1
        a1[2] = rot90<DIR>(a1[2]);         // W8^2 = -j
        a1[3] = mulc<DIR>(a1[3], -h, h);   // W8^3
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            v[k1] = cadd(a0[k1], a1[k1]);
            v[k1 + 4] = csub(a0[k1], a1[k1]);
        }
    }
};

template <int DIR>
struct Butterfly<16, DIR> {
    // n = 4 n1 + n2, k = k1 + 4 k2:  X[k1 + 4 k2] = sum_n2 W4^(n2 k2) W16^(n2 k1) sum_n1 x[4 n1 + n2] W4^(n1 k1)
    template <class C>
    __device__ __forceinline__ static void run(C *v) {
        const float h = 0.70710678118654752440f;
        const float c = 0.92387953251128675613f;  // cos(pi/8)
        const float s = 0.38268343236508977173f;  // sin(pi/8)
        C a[4][4];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            a[n2][0] = v[n2];
            a[n2][1] = v[n2 + 4];
            a[n2][2] = v[n2 + 8];
            a[n2][3] = v[n2 + 12];
            Butterfly<4, DIR>::run(a[n2]);
        }
        // W16^(n2 k1)
        a[1][1] = mulc<DIR>(a[1][1], c, s);    // W16^1
        a[1][2] = mulc<DIR>(a[1][2], h, h);    // W16^2
        a[1][3] = mulc<DIR>(a[1][3], s, c);    // W16^3
        a[2][1] = mulc<DIR>(a[2][1], h, h);    // W16^2
        a[2][2] = rot90<DIR>(a[2][2]);         // W16^4 = -j
        a[2][3] = mulc<DIR>(a[2][3], -h, h);   // W16^6
        a[3][1] = mulc<DIR>(a[3][1], s, c);    // W16^3
        a[3][2] = mulc<DIR>(a[3][2], -h, h);   // W16^6
        a[3][3] = mulc<DIR>(a[3][3], -c, -s);  // W16^9 = -W16^1
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            C u[4] = {a[0][k1], a[1][k1], a[2][k1], a[3][k1]};
            Butterfly<4, DIR>::run(u);
            v[k1] = u[0];
            v[k1 + 4] = u[1];
            v[k1 + 8] = u[2];
            v[k1 + 12] = u[3];
        }
    }
};

// One Stockham stage of radix R over T transforms of length p.S in buf[j*Spad + i].
// buf[j*Spad + lds_phys(i)]; requires (S/R)*T <= floor(16/R) * blockDim.x  (checked on the host).
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
            const float2 *src = buf + jbase[i];
#pragma unroll
            for (int q = 0; q < R; ++q) v[i][q] = src[lds_phys(bb + q * nb)];
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
                // Table gathers are texture-addresser bound (up to 64 cache lines per
                // instruction), so the large radices fetch only the power-of-two multiples
                // and build the rest as products of at most four exact factors.
                if constexpr (R == 16 || R == 8) {
                    float2 w[R];
                    w[1] = p.tw[kt];
                    w[2] = p.tw[2 * kt];
                    w[4] = p.tw[4 * kt];
                    if constexpr (R == 16) w[8] = p.tw[8 * kt];
                    if (DIR > 0) {
                        w[1].y = -w[1].y;
                        w[2].y = -w[2].y;
                        w[4].y = -w[4].y;
                        if constexpr (R == 16) w[8].y = -w[8].y;
                    }
                    w[3] = cmul(w[1], w[2]);
                    w[5] = cmul(w[1], w[4]);
                    w[6] = cmul(w[2], w[4]);
                    w[7] = cmul(w[3], w[4]);
                    if constexpr (R == 16) {
#pragma unroll
                        for (int q = 1; q < 8; ++q) w[8 + q] = cmul(w[q], w[8]);
                    }
#pragma unroll
                    for (int q = 1; q < R; ++q) v[i][q] = cmul(v[i][q], w[q]);
                } else {
#pragma unroll
                    for (int q = 1; q < R; ++q) {
                        float2 w = p.tw[q * kt];
                        if (DIR > 0) w.y = -w.y;
                        v[i][q] = cmul(v[i][q], w);
                    }
                }
            }
            Butterfly<R, DIR>::run(v[i]);
            float2 *dst = buf + jbase[i];
            const int j0 = hi * Ns * R + k;
#pragma unroll
            for (int q = 0; q < R; ++q) dst[lds_phys(j0 + q * Ns)] = v[i][q];
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
            case 5: fft_stage<5, DIR>(buf, p, st, Spad, T, tid, nthr); break;
            case 8: fft_stage<8, DIR>(buf, p, st, Spad, T, tid, nthr); break;
            default: fft_stage<16, DIR>(buf, p, st, Spad, T, tid, nthr); break;
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
