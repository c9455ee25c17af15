// The butterflies of bds_fft_fma.h on packed fp32 pairs (gfx950 v_pk_*_f32: one instruction works on the real and the imaginary
// part of a complex value held in an aligned VGPR pair).
//
// Why: a wave gets a vector instruction issued only every ~4.3 cycles (tools/probe/coissue.hip), a plain fp32 instruction keeps
// the SIMD's vector pipe busy for 2 of them, a packed one for 4 -- the same pipe time per flop, HALF the issue slots.  With the
// two or three waves per SIMD the search kernels can afford, the issue slots are what runs out (vector pipe 43 - 48 % busy in
// round 3), not the pipe.  Every complex operation of a butterfly maps onto packed instructions without extra moves because
// VOP3P source modifiers select (op_sel / op_sel_hi) and negate (neg_lo / neg_hi) the halves of each operand:
//   a + b, a - b                 v_pk_add_f32 (neg on b)                                1 instead of 2
//   a + j b, a - j b             v_pk_add_f32, b's halves swapped, one of them negated    1 instead of 2
//   a w                          v_pk_mul_f32 (a.x a.x)(w.x w.y), v_pk_fma_f32 (a.y a.y)(-w.y w.x) + .    2 instead of 4
//   (a + w b, a - w b)           two v_pk_fma_f32 for the sum, 2 a - sum for the difference   3 instead of 6
// The compiler folds whole-vector negation and broadcasts into these modifiers but not "swap and negate one half", so the
// primitives are inline assembly (plain VALU: no hazards beyond the register dependences the compiler tracks).
// Inverse direction only (DIR = +1: j x = (-x.y, x.x)), which is all the search uses.
#pragma once

#include "bds_fft_fma.h"

namespace bds {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f to_v2f(float2 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ float2 to_f2(v2f a) { return make_float2(a.x, a.y); }

// a + j b
__device__ __forceinline__ v2f pk_addj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a - j b
__device__ __forceinline__ v2f pk_subj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a w
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
// (p, q) = (a + w b, a - w b)
__device__ __forceinline__ void pk_bf2w(v2f a, v2f b, v2f w, v2f &p, v2f &q) {
    v2f t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(b), "v"(w), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(p) : "v"(b), "v"(w), "v"(t));
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(p));
}

// the same with w a compile-time constant: kept in a scalar register pair (the constants of the second butterfly layer would
// otherwise occupy ten VGPRs)
__device__ __forceinline__ v2f pk_cmul_k(v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}
__device__ __forceinline__ void pk_bf2w_k(v2f a, v2f b, v2f w, v2f &p, v2f &q) {
    v2f t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(b), "s"(w), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(p) : "v"(b), "s"(w), "v"(t));
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(p));
}

// inverse 4-point transform in place
__device__ __forceinline__ void pk_radix4(v2f (&v)[4]) {
    const v2f p = v[0] + v[2], q = v[0] - v[2], r = v[1] + v[3], t = v[1] - v[3];
    v[0] = p + r;
    v[1] = pk_addj(q, t);
    v[2] = p - r;
    v[3] = pk_subj(q, t);
}
// inverse 4-point transform over x0, t1 x1, t2 x2, t3 x3
__device__ __forceinline__ void pk_radix4_tw(v2f (&v)[4], v2f t1, v2f t2, v2f t3) {
    v2f p, q, r, s;
    pk_bf2w(v[0], v[2], t2, p, q);
    const v2f x1 = pk_cmul(v[1], t1);
    pk_bf2w(x1, v[3], t3, r, s);
    v[0] = p + r;
    v[1] = pk_addj(q, s);
    v[2] = p - r;
    v[3] = pk_subj(q, s);
}

__device__ __forceinline__ void pk_radix4_tw_k(v2f (&v)[4], v2f t1, v2f t2, v2f t3) {
    v2f p, q, r, s;
    pk_bf2w_k(v[0], v[2], t2, p, q);
    const v2f x1 = pk_cmul_k(v[1], t1);
    pk_bf2w_k(x1, v[3], t3, r, s);
    v[0] = p + r;
    v[1] = pk_addj(q, s);
    v[2] = p - r;
    v[3] = pk_subj(q, s);
}

// 16-point inverse transform in two layers, as bfly16_l1 / bfly16_l2 of bds_fft_fma.h
template <bool TW, bool TW0 = false>
__device__ __forceinline__ void pk_bfly16_l1(const v2f *v, const v2f *tw, int n2, v2f (&a)[4]) {
    a[0] = v[n2];
    a[1] = v[n2 + 4];
    a[2] = v[n2 + 8];
    a[3] = v[n2 + 12];
    if constexpr (TW) {
        if (n2 > 0 || TW0) a[0] = pk_cmul(a[0], tw[n2]);
        pk_radix4_tw(a, tw[n2 + 4], tw[n2 + 8], tw[n2 + 12]);
    } else {
        pk_radix4(a);
    }
}
template <int K1>
__device__ __forceinline__ void pk_bfly16_l2(const v2f (&a)[4][4], v2f (&u)[4]) {
    const float h = 0.70710678118654752440f;
    const float c = 0.92387953251128675613f;  // cos(pi/8)
    const float s = 0.38268343236508977173f;  // sin(pi/8)
    if constexpr (K1 == 0) {
        u[0] = a[0][0], u[1] = a[1][0], u[2] = a[2][0], u[3] = a[3][0];
        pk_radix4(u);
    } else if constexpr (K1 == 1) {
        u[0] = a[0][1], u[1] = a[1][1], u[2] = a[2][1], u[3] = a[3][1];
        pk_radix4_tw_k(u, (v2f){c, s}, (v2f){h, h}, (v2f){s, c});  // W16^1, W16^2, W16^3
    } else if constexpr (K1 == 2) {  // W16^2, W16^4 = j, W16^6
        const v2f p = pk_addj(a[0][2], a[2][2]), q = pk_subj(a[0][2], a[2][2]);
        const v2f x1 = pk_cmul_k(a[1][2], (v2f){h, h});
        v2f r, t;
        pk_bf2w_k(x1, a[3][2], (v2f){-h, h}, r, t);
        u[0] = p + r, u[1] = pk_addj(q, t), u[2] = p - r, u[3] = pk_subj(q, t);
    } else {
        u[0] = a[0][3], u[1] = a[1][3], u[2] = a[2][3], u[3] = a[3][3];
        pk_radix4_tw_k(u, (v2f){s, c}, (v2f){-h, h}, (v2f){-c, -s});  // W16^3, W16^6, W16^9
    }
}
// v <- DFT16(tw .* v), inverse
template <bool TW, bool TW0 = false>
__device__ __forceinline__ void pk_bfly16(v2f *v, const v2f *tw) {
    v2f a[4][4], u[4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) pk_bfly16_l1<TW, TW0>(v, tw, n2, a[n2]);
    pk_bfly16_l2<0>(a, u);
    v[0] = u[0], v[4] = u[1], v[8] = u[2], v[12] = u[3];
    pk_bfly16_l2<1>(a, u);
    v[1] = u[0], v[5] = u[1], v[9] = u[2], v[13] = u[3];
    pk_bfly16_l2<2>(a, u);
    v[2] = u[0], v[6] = u[1], v[10] = u[2], v[14] = u[3];
    pk_bfly16_l2<3>(a, u);
    v[3] = u[0], v[7] = u[1], v[11] = u[2], v[15] = u[3];
}

// 8-point inverse transform: layer 1 = DFT4 over the even / odd inputs, layer 2 group k1: (v[k1], v[k1 + 4]) = a0[k1] +- W8^k1 a1[k1]
template <bool TW>
__device__ __forceinline__ void pk_bfly8_l1(const v2f *v, const v2f *tw, int odd, v2f (&a)[4]) {
    a[0] = v[odd], a[1] = v[2 + odd], a[2] = v[4 + odd], a[3] = v[6 + odd];
    if constexpr (TW) {
        if (odd) a[0] = pk_cmul(a[0], tw[1]);
        pk_radix4_tw(a, tw[2 + odd], tw[4 + odd], tw[6 + odd]);
    } else {
        pk_radix4(a);
    }
}
template <int K1>
__device__ __forceinline__ void pk_bfly8_l2(const v2f (&a0)[4], const v2f (&a1)[4], v2f &lo, v2f &hi) {
    const float h = 0.70710678118654752440f;
    if constexpr (K1 == 0) {
        lo = a0[0] + a1[0];
        hi = a0[0] - a1[0];
    } else if constexpr (K1 == 1) {
        pk_bf2w_k(a0[1], a1[1], (v2f){h, h}, lo, hi);
    } else if constexpr (K1 == 2) {
        lo = pk_addj(a0[2], a1[2]);
        hi = pk_subj(a0[2], a1[2]);
    } else {
        pk_bf2w_k(a0[3], a1[3], (v2f){-h, h}, lo, hi);
    }
}
template <bool TW>
__device__ __forceinline__ void pk_bfly8(v2f *v, const v2f *tw) {
    v2f a0[4], a1[4];
    pk_bfly8_l1<TW>(v, tw, 0, a0);
    pk_bfly8_l1<TW>(v, tw, 1, a1);
    pk_bfly8_l2<0>(a0, a1, v[0], v[4]);
    pk_bfly8_l2<1>(a0, a1, v[1], v[5]);
    pk_bfly8_l2<2>(a0, a1, v[2], v[6]);
    pk_bfly8_l2<3>(a0, a1, v[3], v[7]);
}

// 3-point inverse transform: X0 = a + t, X1,2 = (a - t / 2) +- j s d  with t = b + c, d = b - c, s = sin(2 pi / 3)     (6 instead of 14)
__device__ __forceinline__ void pk_radix3(v2f &x0, v2f &x1, v2f &x2) {
    const v2f ks = {0.86602540378443864676f, 0.86602540378443864676f};
    const v2f t = x1 + x2, d = x1 - x2;
    v2f m;
    asm("v_pk_fma_f32 %0, %1, -0.5, %2 op_sel_hi:[1,0,1]" : "=v"(m) : "v"(t), "v"(x0));
    x0 = x0 + t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(x1) : "v"(d), "s"(ks), "v"(m));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(x2) : "v"(d), "s"(ks), "v"(m));
}
// 12-point inverse transform (n = 3 a + b, k = c + 4 d, as Butterfly<12, +1> of bds_acq_wcols.h)
__device__ __forceinline__ void pk_bfly12(v2f *v) {
    const float h = 0.5f, s = 0.86602540378443864676f;
    v2f t[3][4];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[b][0] = v[b], t[b][1] = v[b + 3], t[b][2] = v[b + 6], t[b][3] = v[b + 9];
        pk_radix4(t[b]);
    }
    t[1][1] = pk_cmul_k(t[1][1], (v2f){s, h});   // W12^1
    t[1][2] = pk_cmul_k(t[1][2], (v2f){h, s});   // W12^2
    t[2][1] = pk_cmul_k(t[2][1], (v2f){h, s});   // W12^2
    t[2][2] = pk_cmul_k(t[2][2], (v2f){-h, s});  // W12^4
    // W12^3 = j on t[1][3], W12^6 = -1 on t[2][3]: folded into the 3-point transform of c = 3 below
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pk_radix3(t[0][c], t[1][c], t[2][c]);
        v[c] = t[0][c], v[c + 4] = t[1][c], v[c + 8] = t[2][c];
    }
    {   // c = 3: inputs a = t0, b = j t1, c = -t2:  t = j t1 - t2, d = j t1 + t2
        const v2f ks = {s, s};
        const v2f a = t[0][3];
        v2f tt, d, m;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,1] neg_hi:[0,1]" : "=v"(tt) : "v"(t[1][3]), "v"(t[2][3]));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(d) : "v"(t[1][3]), "v"(t[2][3]));
        asm("v_pk_fma_f32 %0, %1, -0.5, %2 op_sel_hi:[1,0,1]" : "=v"(m) : "v"(tt), "v"(a));
        v[3] = a + tt;
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(v[7]) : "v"(d), "s"(ks), "v"(m));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(v[11]) : "v"(d), "s"(ks), "v"(m));
    }
}

// a w with w in a scalar register pair (a workgroup-uniform factor)
__device__ __forceinline__ v2f pk_cmul_s(v2f a, v2f w) { return pk_cmul_k(a, w); }

// ---- one spelling for both representations (float2: bds_fft_fma.h, v2f: packed), inverse direction ------------------------
__device__ __forceinline__ void cx_set(float2 &d, float x, float y) { d = make_float2(x, y); }
__device__ __forceinline__ void cx_set(v2f &d, float x, float y) { d = (v2f){x, y}; }
__device__ __forceinline__ float2 cx_mul(float2 a, float2 b) { return cmul(a, b); }
__device__ __forceinline__ v2f cx_mul(v2f a, v2f b) { return pk_cmul(a, b); }
__device__ __forceinline__ float2 cx_f2(float2 a) { return a; }
__device__ __forceinline__ float2 cx_f2(v2f a) { return make_float2(a.x, a.y); }
// product with a workgroup-uniform factor held in scalar registers
__device__ __forceinline__ float2 cx_mul_uniform(float2 a, float sx, float sy) {
    return make_float2(fmaf(-a.y, sy, a.x * sx), fmaf(a.y, sx, a.x * sy));
}
__device__ __forceinline__ v2f cx_mul_uniform(v2f a, float sx, float sy) { return pk_cmul_s(a, (v2f){sx, sy}); }
// untwiddled R-point inverse transform (first stage of the column pass)
template <int R>
__device__ __forceinline__ void cx_bfly(float2 *v) {
    Butterfly<R, +1>::run(v);
}
template <int R>
__device__ __forceinline__ void cx_bfly(v2f *v) {
    static_assert(R == 4 || R == 8 || R == 12 || R == 16, "radix of the column pass's first stage");
    if constexpr (R == 4) {
        v2f a[4] = {v[0], v[1], v[2], v[3]};
        pk_radix4(a);
        v[0] = a[0], v[1] = a[1], v[2] = a[2], v[3] = a[3];
    } else if constexpr (R == 8) {
        pk_bfly8<false>(v, nullptr);
    } else if constexpr (R == 12) {
        pk_bfly12(v);
    } else {
        pk_bfly16<false>(v, nullptr);
    }
}
template <bool TW>
__device__ __forceinline__ void cx_bfly8(float2 *v, const float2 *tw) {
    bfly8_fma<+1, TW>(v, tw);
}
template <bool TW>
__device__ __forceinline__ void cx_bfly8(v2f *v, const v2f *tw) {
    pk_bfly8<TW>(v, tw);
}
template <bool TW, bool TW0 = false>
__device__ __forceinline__ void cx_bfly16_l1(const float2 *v, const float2 *tw, int n2, float2 (&a)[4]) {
    bfly16_l1<+1, TW, TW0>(v, tw, n2, a);
}
template <bool TW, bool TW0 = false>
__device__ __forceinline__ void cx_bfly16_l1(const v2f *v, const v2f *tw, int n2, v2f (&a)[4]) {
    pk_bfly16_l1<TW, TW0>(v, tw, n2, a);
}
template <int K1>
__device__ __forceinline__ void cx_bfly16_l2(const float2 (&a)[4][4], float2 (&u)[4]) {
    bfly16_l2<+1, K1>(a, u);
}
template <int K1>
__device__ __forceinline__ void cx_bfly16_l2(const v2f (&a)[4][4], v2f (&u)[4]) {
    pk_bfly16_l2<K1>(a, u);
}

}  // namespace bds
