/* CPU oracle, C restatement of ONE tracking epoch's correlators (TEST INFRASTRUCTURE ONLY).
 *
 * float64 restatement of the sample loop of
 *   B2a/tracking.m:260-331          code indices ceil(tcode)+1 into [c(end) c c(1)], carrier exp(+j trigarg), I = imag, Q = real
 *   B1C/NB_tracking.m:271-345       half-chip BOC(1,1) codes, carrier exp(-j trigarg), I = real, Q = imag
 *   B1C/WB_tracking.m:289-372       ... and the BOC(6,1) pilot replica indexed ceil(tcode * 6) + 1
 * -- what oracle/tracking.py does per epoch with NumPy vectors, here as one scalar loop (the loop filters, the discriminators
 * and the C/N0 estimator stay in oracle/tracking.py, which calls this through its `correlate=` hook).  Written from the .m
 * lines, sharing no code with the NumPy statements it replaces: the C library's sin / cos, a plain running sum.  It makes the
 * oracle fast enough for BASELINE.json configs[3] at full size (12 channels x 3 600 epochs x 993 750 samples: ~1.5 h of NumPy,
 * minutes here with one thread per channel).  tests/test_oracle_c.py holds it against the NumPy epoch.
 *
 * PARITY UNPINNED like the rest of oracle/.  Only tests/ may load it; the product library never does.
 */
#define _GNU_SOURCE /* M_PI under -std=c11 */
#include <math.h>
#include <stdint.h>

/* raw       : the epoch's samples as read from the record (int8; interleaved I/Q pairs when iq)
 * blk       : samples of the epoch (blksize, tracking.m:233)
 * rem_code, step, spc_el : remCodePhase, codePhaseStep, earlyLateSpc (chips)
 * scale     : 1 (B2a: chips) or 2 (B1C: half chips, "tcode * 2", NB_tracking.m:271-306)
 * rem_carr, carr_freq, fs : remCarrPhase, carrFreq, samplingFreq
 * b2a       : 1 = exp(+j trigarg), I = imag, Q = real (tracking.m:309-314); 0 = exp(-j trigarg), I = real, Q = imag (NB:320-325)
 * dcode / pcode / p6code : the extended code arrays [c(end) c c(1)] as doubles; pcode / p6code may be NULL
 * sums[18]  : I_E Q_E I_P Q_P I_L Q_L of the data code, of the pilot BOC(1,1) / B2a pilot code, of the pilot BOC(6,1)
 * t_p_last  : tcode of the prompt replica at the last sample (tracking.m:295 / WB:327 derive the next remCodePhase from it)
 * trig_end  : trigarg(blksize + 1) (tracking.m:304-305: remCarrPhase = rem(trigarg(blksize+1), 2 pi)) */
/* MATLAB's a:d:b for the non-integer operands of the tcode vectors (tracking.m:260-262, NB_tracking.m:271-273, WB_tracking.m:289-291),
 * after the algorithm MathWorks published for the built-in (colonop.m, Technical Solution 1-4FLI96): n = round((b-a)/d) intervals
 * (one less if a+n*d overshoots b by more than tol = 2 eps max(|a|,|b|)), right end c = a+n*d snapped to b within tol, elements
 * 0..floor(n/2) = a + k d, elements n-floor(n/2)..n = c - (n-k) d, mid-point of an even n = (a+c)/2.  Written from that description,
 * independently of oracle/matlab.py (tests/test_oracle_c.py holds the two together). */
typedef struct { double a, d, c; long n, h; int even; } colon_t;
static int colon_init(colon_t *v, double a, double d, double b) {
    const double tol = 2.0 * 2.220446049250313e-16 * fmax(fabs(a), fabs(b));
    if (!(d > 0.0) || b < a) return -1; /* the tcode vectors ascend */
    double q = (b - a) / d;
    long n = (long)floor(fabs(q) + 0.5); /* MATLAB round(): half away from zero; q >= 0 here */
    if (a + (double)n * d - b > tol) n -= 1;
    double c = a + (double)n * d;
    if (c - b > -tol) c = b;
    v->a = a, v->d = d, v->c = c, v->n = n, v->h = n / 2, v->even = (n % 2 == 0);
    return 0;
}
/* diagnostics only (tools/colon_effect.py): 1 = every element as a + k d, the form rounds 1-5 of this repo used */
static int g_plain_colon = 0;
void bds_oracle_trk_set_plain_colon(int on) { g_plain_colon = on; }
static inline double colon_at(const colon_t *v, long k) {
    if (g_plain_colon) return v->a + (double)k * v->d;
    if (k > v->h) return v->c - (double)(v->n - k) * v->d;
    if (v->even && k == v->h) return (v->a + v->c) / 2.0;
    return v->a + (double)k * v->d;
}

int bds_oracle_trk_epoch(const int8_t *raw, long blk, int iq, double rem_code, double step, double spc_el, double scale, double rem_carr,
                         double carr_freq, double fs, int b2a, const double *dcode, const double *pcode, const double *p6code, double *sums,
                         double *t_p_last, double *trig_end) {
    if (!raw || !dcode || !sums || blk < 1) return -1;
    const double inc = step * scale;
    const double last = (double)(blk - 1) * step + rem_code; /* (blksize-1)*codePhaseStep + remCodePhase, then -+ earlyLateSpc, then *2 */
    colon_t ve, vl, vp;
    if (colon_init(&ve, (rem_code - spc_el) * scale, inc, (last - spc_el) * scale) || colon_init(&vl, (rem_code + spc_el) * scale, inc, (last + spc_el) * scale) ||
        colon_init(&vp, rem_code * scale, inc, last * scale))
        return -2;
    if (ve.n != blk - 1 || vl.n != blk - 1 || vp.n != blk - 1) return -3; /* MATLAB would stop at tcode(blksize) / the .* products */
    const double w = carr_freq * 2.0 * M_PI; /* (carrFreq * 2.0 * pi) .* time  -- tracking.m:304 */
    double acc[18];
    for (int i = 0; i < 18; ++i) acc[i] = 0.0;
    double tp = vp.a;
    for (long k = 0; k < blk; ++k) {
        const double kk = (double)k;
        const double te = colon_at(&ve, k), tl = colon_at(&vl, k);
        tp = colon_at(&vp, k);
        const long ie = (long)ceil(te), il = (long)ceil(tl), ip = (long)ceil(tp); /* 0-based = MATLAB's ceil(t) + 1 */
        const double trig = w * (kk / fs) + rem_carr;
        const double c = cos(trig), s = sin(trig);
        double re, im;
        if (iq)
            re = raw[2 * k], im = raw[2 * k + 1];
        else
            re = raw[k], im = 0.0;
        double i_bb, q_bb;
        if (b2a) { /* (c + j s)(re + j im): Q = real, I = imag */
            q_bb = c * re - s * im;
            i_bb = c * im + s * re;
        } else { /* (c - j s)(re + j im): I = real, Q = imag */
            i_bb = c * re + s * im;
            q_bb = c * im - s * re;
        }
        const double de = dcode[ie], dp = dcode[ip], dl = dcode[il];
        acc[0] += de * i_bb, acc[1] += de * q_bb, acc[2] += dp * i_bb, acc[3] += dp * q_bb, acc[4] += dl * i_bb, acc[5] += dl * q_bb;
        if (pcode) {
            const double pe = pcode[ie], pp = pcode[ip], pl = pcode[il];
            acc[6] += pe * i_bb, acc[7] += pe * q_bb, acc[8] += pp * i_bb, acc[9] += pp * q_bb, acc[10] += pl * i_bb, acc[11] += pl * q_bb;
        }
        if (p6code) { /* WB:298,311,324: ceil(tcode * 6) + 1 */
            const double se = p6code[(long)ceil(te * 6.0)], sp = p6code[(long)ceil(tp * 6.0)], sl = p6code[(long)ceil(tl * 6.0)];
            acc[12] += se * i_bb, acc[13] += se * q_bb, acc[14] += sp * i_bb, acc[15] += sp * q_bb, acc[16] += sl * i_bb, acc[17] += sl * q_bb;
        }
    }
    for (int i = 0; i < 18; ++i) sums[i] = acc[i];
    if (t_p_last) *t_p_last = tp;
    if (trig_end) *trig_end = w * ((double)blk / fs) + rem_carr;
    return 0;
}

/* diagnostics (tools/colon_effect.py): samples of one epoch whose replica index differs between MATLAB's colon vector and a + k d.
 * counts[0..2] = E, P, L code index ceil(tcode); counts[3..5] = E, P, L BOC(6,1) index ceil(tcode * 6); max_ulp = the largest
 * |difference| of the two tcode values in units of the spacing of doubles at tcode. */
int bds_oracle_trk_colon_diff(long blk, double rem_code, double step, double spc_el, double scale, long *counts, double *max_ulp) {
    const double inc = step * scale;
    const double last = (double)(blk - 1) * step + rem_code;
    colon_t v[3];
    if (colon_init(&v[0], (rem_code - spc_el) * scale, inc, (last - spc_el) * scale) || colon_init(&v[1], rem_code * scale, inc, last * scale) ||
        colon_init(&v[2], (rem_code + spc_el) * scale, inc, (last + spc_el) * scale))
        return -2;
    const int keep = g_plain_colon;
    g_plain_colon = 0;
    double mu = 0.0;
    for (int i = 0; i < 6; ++i) counts[i] = 0;
    for (int r = 0; r < 3; ++r) {
        if (v[r].n != blk - 1) { g_plain_colon = keep; return -3; }
        for (long k = 0; k < blk; ++k) {
            const double tm = colon_at(&v[r], k), tp = v[r].a + (double)k * v[r].d;
            if (ceil(tm) != ceil(tp)) counts[r] += 1;
            if (ceil(tm * 6.0) != ceil(tp * 6.0)) counts[3 + r] += 1;
            if (tm != tp) {
                const double u = fabs(tm - tp) / (nextafter(fabs(tm), INFINITY) - fabs(tm));
                if (u > mu) mu = u;
            }
        }
    }
    g_plain_colon = keep;
    if (max_ulp) *max_ulp = mu;
    return 0;
}
