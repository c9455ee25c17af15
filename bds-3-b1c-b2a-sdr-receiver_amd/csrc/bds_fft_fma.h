// Radix-16 butterfly with the twiddles folded into fused multiply-adds (gfx950: an fp32 add, a multiply and an FMA all cost one
// 2-cycle issue slot, so what counts is the number of instructions, not of flops).
//
//   (a + w b, a - w b):  p = fma(w.x, b.x, fma(-w.y, b.y, a.x)), ...;  q = 2 a - p        6 instructions instead of 4 + 4
//
// Applied (1) to the constant twiddles W16^(n2 k1) between the two radix-4 layers (86 instead of 96 instructions for that layer)
// and (2) to INPUT twiddles tw[1 .. 15] of the transform (a stage twiddle applied to what a lane has just read, instead of to what
// it is about to write): the first layer takes 108 instead of 60 + 64 instructions.  A twiddled 16-point transform: 194 against
// 220; an untwiddled one: 150 against 160.  Used by the wave-private row pass (bds_acq_wrows.h) and, as the 8-point version,
// by the wave-private column pass (bds_acq_wcols.h).
#pragma once

#include "bds_fft.h"

namespace bds {

// (p, q) = (a + w b, a - w b)
__device__ __forceinline__ void bf2w(float2 a, float2 b, float wr, float wi, float2 &p, float2 &q) {
    p.x = fmaf(wr, b.x, fmaf(-wi, b.y, a.x));
    p.y = fmaf(wr, b.y, fmaf(wi, b.x, a.y));
    q.x = fmaf(2.f, a.x, -p.x);
    q.y = fmaf(2.f, a.y, -p.y);
}

// radix 4 over x0, t1 x1, t2 x2, t3 x3 (x0 as it is)
template <int DIR>
__device__ __forceinline__ void radix4_tw(float2 *v, float2 t1, float2 t2, float2 t3) {
    float2 p, q, r, s;
    bf2w(v[0], v[2], t2.x, t2.y, p, q);
    const float2 x1 = cmul(v[1], t1);
    bf2w(x1, v[3], t3.x, t3.y, r, s);
    s = rot90<DIR>(s);
    v[0] = cadd(p, r);
    v[1] = cadd(q, s);
    v[2] = csub(p, r);
    v[3] = csub(q, s);
}

// The 16-point transform in two layers, so that a caller can put LDS or global accesses between the groups of a layer (the
// first group of layer 1 needs only inputs 0, 4, 8, 12; group k1 of layer 2 delivers outputs k1, k1 + 4, k1 + 8, k1 + 12):
//   layer 1, group n2: a[n2][.] = DFT4 over (tw .* v)[n2 + 4 m]        (TW: tw[1 .. 15] on the inputs, tw[0] only if TW0)
//   layer 2, group k1: v[k1 + 4 k2] = DFT4 over W16^(n2 k1) a[n2][k1]
template <int DIR, bool TW, bool TW0 = false>
__device__ __forceinline__ void bfly16_l1(const float2 *v, const float2 *tw, int n2, float2 (&a)[4]) {
    a[0] = v[n2];
    a[1] = v[n2 + 4];
    a[2] = v[n2 + 8];
    a[3] = v[n2 + 12];
    if constexpr (TW) {
        if (n2 > 0 || TW0) a[0] = cmul(a[0], tw[n2]);
        radix4_tw<DIR>(a, tw[n2 + 4], tw[n2 + 8], tw[n2 + 12]);
    } else {
        Butterfly<4, DIR>::run(a);
    }
}
template <int DIR, int K1>
__device__ __forceinline__ void bfly16_l2(const float2 (&a)[4][4], float2 (&u)[4]) {
    constexpr float sg = DIR > 0 ? 1.f : -1.f;
    const float h = 0.70710678118654752440f;
    const float c = 0.92387953251128675613f;  // cos(pi/8)
    const float s = 0.38268343236508977173f;  // sin(pi/8)
    if constexpr (K1 == 0) {
        u[0] = a[0][0], u[1] = a[1][0], u[2] = a[2][0], u[3] = a[3][0];
        Butterfly<4, DIR>::run(u);
    } else if constexpr (K1 == 1) {
        u[0] = a[0][1], u[1] = a[1][1], u[2] = a[2][1], u[3] = a[3][1];
        radix4_tw<DIR>(u, make_float2(c, sg * s), make_float2(h, sg * h), make_float2(s, sg * c));  // W16^1, W16^2, W16^3
    } else if constexpr (K1 == 2) {  // W16^2, W16^4 = +-j, W16^6
        const float2 x0 = a[0][2], jx2 = rot90<DIR>(a[2][2]);
        const float2 p = cadd(x0, jx2), q = csub(x0, jx2);
        const float2 x1 = cmul(a[1][2], make_float2(h, sg * h));
        float2 r, t;
        bf2w(x1, a[3][2], -h, sg * h, r, t);
        t = rot90<DIR>(t);
        u[0] = cadd(p, r), u[1] = cadd(q, t), u[2] = csub(p, r), u[3] = csub(q, t);
    } else {
        u[0] = a[0][3], u[1] = a[1][3], u[2] = a[2][3], u[3] = a[3][3];
        radix4_tw<DIR>(u, make_float2(s, sg * c), make_float2(-h, sg * h), make_float2(-c, -sg * s));  // W16^3, W16^6, W16^9
    }
}

// v <- DFT16(tw .* v) (TW: tw[1 .. 15] are applied to the inputs, tw[0] is taken as 1 unless TW0) or DFT16(v); DIR as Butterfly<16, DIR>
template <int DIR, bool TW, bool TW0 = false>
__device__ __forceinline__ void bfly16_fma(float2 *v, const float2 *tw) {
#ifdef BDS_EXP_PLAIN_BFLY  // timing experiment: the same transform with separate twiddle products and Butterfly<16, DIR>
    if constexpr (TW) {
#pragma unroll
        for (int i = TW0 ? 0 : 1; i < 16; ++i) v[i] = cmul(v[i], tw[i]);
    }
    Butterfly<16, DIR>::run(v);
    return;
#endif
    float2 a[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) bfly16_l1<DIR, TW, TW0>(v, tw, n2, a[n2]);
    float2 u[4];
    bfly16_l2<DIR, 0>(a, u);
    v[0] = u[0], v[4] = u[1], v[8] = u[2], v[12] = u[3];
    bfly16_l2<DIR, 1>(a, u);
    v[1] = u[0], v[5] = u[1], v[9] = u[2], v[13] = u[3];
    bfly16_l2<DIR, 2>(a, u);
    v[2] = u[0], v[6] = u[1], v[10] = u[2], v[14] = u[3];
    bfly16_l2<DIR, 3>(a, u);
    v[3] = u[0], v[7] = u[1], v[11] = u[2], v[15] = u[3];
}

// The 8-point transform in two layers (as the 16-point one above): layer 1 = DFT4 over the even inputs (a0) and over the odd
// inputs (a1), each with its input twiddles; layer 2, group k1: (v[k1], v[k1 + 4]) = a0[k1] +- W8^k1 a1[k1]
template <int DIR, bool TW>
__device__ __forceinline__ void bfly8_l1(const float2 *v, const float2 *tw, int odd, float2 (&a)[4]) {
    a[0] = v[odd], a[1] = v[2 + odd], a[2] = v[4 + odd], a[3] = v[6 + odd];
    if constexpr (TW) {
        if (odd) a[0] = cmul(a[0], tw[1]);
        radix4_tw<DIR>(a, tw[2 + odd], tw[4 + odd], tw[6 + odd]);
    } else {
        Butterfly<4, DIR>::run(a);
    }
}
template <int DIR, int K1>
__device__ __forceinline__ void bfly8_l2(const float2 (&a0)[4], const float2 (&a1)[4], float2 &lo, float2 &hi) {
    constexpr float sg = DIR > 0 ? 1.f : -1.f;
    const float h = 0.70710678118654752440f;
    if constexpr (K1 == 0) {
        lo = cadd(a0[0], a1[0]);
        hi = csub(a0[0], a1[0]);
    } else if constexpr (K1 == 1) {
        bf2w(a0[1], a1[1], h, sg * h, lo, hi);
    } else if constexpr (K1 == 2) {
        const float2 j2 = rot90<DIR>(a1[2]);
        lo = cadd(a0[2], j2);
        hi = csub(a0[2], j2);
    } else {
        bf2w(a0[3], a1[3], -h, sg * h, lo, hi);
    }
}

// v <- DFT8(tw .* v) (TW: tw[1 .. 7] on the inputs) or DFT8(v): 72 instructions against 28 + 56, 52 against 56
template <int DIR, bool TW>
__device__ __forceinline__ void bfly8_fma(float2 *v, const float2 *tw) {
#ifdef BDS_EXP_PLAIN_BFLY
    if constexpr (TW) {
#pragma unroll
        for (int i = 1; i < 8; ++i) v[i] = cmul(v[i], tw[i]);
    }
    Butterfly<8, DIR>::run(v);
    return;
#endif
    float2 a0[4], a1[4];
    bfly8_l1<DIR, TW>(v, tw, 0, a0);
    bfly8_l1<DIR, TW>(v, tw, 1, a1);
    bfly8_l2<DIR, 0>(a0, a1, v[0], v[4]);
    bfly8_l2<DIR, 1>(a0, a1, v[1], v[5]);
    bfly8_l2<DIR, 2>(a0, a1, v[2], v[6]);
    bfly8_l2<DIR, 3>(a0, a1, v[3], v[7]);
}

}  // namespace bds
