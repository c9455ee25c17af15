/* CPU oracle, C restatement of the coarse search (TEST INFRASTRUCTURE ONLY).
 *
 * float64 / complex128 restatement of the Doppler-row loop of
 *   B1C/acquisition.m:191-222   results(b,:) = (sqrt(11) abs(ifft(fft(carr.*x).*conj(fft(c_d)))) + sqrt(29) abs(ifft(... c_p))) / sqrt(40)
 *   B2a/acquisition.m:187-211   results(b,:) = abs(ifft(fft(carr.*x).*conj(fft(c_d)))) + abs(ifft(... c_p))
 * and of the reductions their callers take from the D x N matrix (B1C :229-232, B2a :218-221: the maximum of every
 * row with its first index, and the running column maximum), written from the .m files like oracle/acquisition.py
 * (which it is checked against in tests/test_oracle_c.py) but sharing no code with it: its own mixed-radix transform
 * (any length; a direct DFT for prime factors such as the 53 in N = 1 987 500), the C library's sin / cos.
 * It exists (SURVEY.md 8d "CPU baseline") as the compiled float64 host implementation the NumPy figures stand beside in
 * bench.py's cpu_baseline leg, and as a second, independent restatement for the parity tests at sizes NumPy needs an
 * hour for.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product library
 * never does, and has no CPU path.
 *
 * PARITY UNPINNED like the rest of oracle/: the reference is MATLAB only and cannot run here.
 *
 * MATLAB's fft is unnormalised, ifft carries 1/N (FFTW conventions): kept.
 */
#define _GNU_SOURCE /* M_PI under -std=c11 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    double re, im;
} cpx;

static inline int omp_tid(void) {
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
}
static inline int omp_nthr(void) {
#ifdef _OPENMP
    return omp_get_max_threads() > 0 ? omp_get_max_threads() : 1;
#else
    return 1;
#endif
}

/* ---- plan: Stockham autosort, decimation in frequency, radices 4 / 2 / 3 / 5 and a direct DFT for any other prime ---- */
#define MAX_STAGES 64
typedef struct {
    long n;
    int nst;
    int radix[MAX_STAGES];
    cpx *tw[MAX_STAGES];  /* stage i: w_n_i^(p k), [p][k = 1 .. r-1], n_i = length at that stage */
    cpx *dft[MAX_STAGES]; /* generic stage: w_r^(i k), [k][i] */
    double *scratch[MAX_STAGES]; /* generic stage: per-thread blocks S, D, R, I */
} plan_t;

static void plan_free(plan_t *p) {
    if (!p) return;
    for (int i = 0; i < p->nst; ++i) {
        free(p->tw[i]);
        free(p->dft[i]);
        free(p->scratch[i]);
    }
    free(p);
}

static plan_t *plan_make(long n) {
    if (n < 1) return NULL;
    plan_t *p = (plan_t *)calloc(1, sizeof(plan_t));
    if (!p) return NULL;
    p->n = n;
    long m = n;
    int f[MAX_STAGES], nf = 0;
    while (m % 4 == 0 && nf < MAX_STAGES) f[nf++] = 4, m /= 4;
    while (m % 2 == 0 && nf < MAX_STAGES) f[nf++] = 2, m /= 2;
    while (m % 3 == 0 && nf < MAX_STAGES) f[nf++] = 3, m /= 3;
    while (m % 5 == 0 && nf < MAX_STAGES) f[nf++] = 5, m /= 5;
    for (long q = 7; m > 1 && nf < MAX_STAGES; q += 2) {
        if (q * q > m) q = m;
        while (m % q == 0 && nf < MAX_STAGES) f[nf++] = (int)q, m /= q;
    }
    if (m != 1) {
        free(p);
        return NULL;
    }
    /* large primes last: their stage then runs with the longest unit-stride inner loop and no twiddles */
    for (int i = 0; i < nf; ++i)
        for (int j = i + 1; j < nf; ++j)
            if (f[j] < f[i]) {
                int t = f[i];
                f[i] = f[j];
                f[j] = t;
            }
    p->nst = nf;
    long len = n;
    for (int i = 0; i < nf; ++i) {
        const int r = f[i];
        const long mm = len / r;
        p->radix[i] = r;
        p->tw[i] = (cpx *)malloc(sizeof(cpx) * (size_t)mm * (size_t)(r - 1 > 0 ? r - 1 : 1));
        if (!p->tw[i]) {
            plan_free(p);
            return NULL;
        }
        for (long q = 0; q < mm; ++q)
            for (int k = 1; k < r; ++k) {
                /* exp(-2 pi i q k / len), argument reduced exactly in integers */
                const long num = (q * k) % len;
                const double a = -2.0 * M_PI * (double)num / (double)len;
                p->tw[i][q * (r - 1) + (k - 1)].re = cos(a);
                p->tw[i][q * (r - 1) + (k - 1)].im = sin(a);
            }
        if (r > 5) {
            p->dft[i] = (cpx *)malloc(sizeof(cpx) * (size_t)r * r);
            p->scratch[i] = (double *)malloc(sizeof(double) * (size_t)omp_nthr() * (size_t)(r + 1) * 2 * 128);
            if (!p->dft[i] || !p->scratch[i]) {
                plan_free(p);
                return NULL;
            }
            for (int k = 0; k < r; ++k)
                for (int j = 0; j < r; ++j) {
                    const double a = -2.0 * M_PI * (double)((k * j) % r) / (double)r;
                    p->dft[i][k * r + j].re = cos(a);
                    p->dft[i][k * r + j].im = sin(a);
                }
        }
        len = mm;
    }
    return p;
}

static inline cpx cmul(cpx a, cpx b) {
    cpx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    return r;
}
static inline cpx cmulc(cpx a, cpx b, int conj_b) { /* a * (conj_b ? conj(b) : b) */
    const double bi = conj_b ? -b.im : b.im;
    cpx r = {a.re * b.re - a.im * bi, a.re * bi + a.im * b.re};
    return r;
}

/* one stage: length len = r m at stride s.  y[q + s (r p + k)] = w_len^(p k) sum_i x[q + s (p + m i)] w_r^(i k).
 * inv: conjugated twiddles (the inverse transform without its 1/N). */
static void stage(const plan_t *P, int st, long len, long s, const cpx *x, cpx *y, int inv) {
    const int r = P->radix[st];
    const long m = len / r;
    const cpx *tw = P->tw[st];
    const double sg = inv ? -1.0 : 1.0; /* sign of the imaginary unit in the butterflies */
    for (long p = 0; p < m; ++p) {
        const cpx *w = tw + p * (r - 1);
        const cpx *xp = x + s * p;
        cpx *yp = y + s * r * p;
        if (r == 2) {
            for (long q = 0; q < s; ++q) {
                const cpx a = xp[q], b = xp[q + s * m];
                cpx d = {a.re - b.re, a.im - b.im};
                yp[q].re = a.re + b.re;
                yp[q].im = a.im + b.im;
                yp[q + s] = cmulc(d, w[0], inv);
            }
        } else if (r == 4) {
            for (long q = 0; q < s; ++q) {
                const cpx a = xp[q], b = xp[q + s * m], c = xp[q + 2 * s * m], d = xp[q + 3 * s * m];
                const cpx t0 = {a.re + c.re, a.im + c.im}, t1 = {a.re - c.re, a.im - c.im};
                const cpx t2 = {b.re + d.re, b.im + d.im};
                /* -j (b - d) forward, +j (b - d) inverse */
                const cpx t3 = {sg * (b.im - d.im), -sg * (b.re - d.re)};
                cpx o1 = {t1.re + t3.re, t1.im + t3.im}, o2 = {t0.re - t2.re, t0.im - t2.im}, o3 = {t1.re - t3.re, t1.im - t3.im};
                yp[q].re = t0.re + t2.re;
                yp[q].im = t0.im + t2.im;
                yp[q + s] = cmulc(o1, w[0], inv);
                yp[q + 2 * s] = cmulc(o2, w[1], inv);
                yp[q + 3 * s] = cmulc(o3, w[2], inv);
            }
        } else if (r == 3) {
            const double h = 0.5, c3 = 0.86602540378443864676 * sg;
            for (long q = 0; q < s; ++q) {
                const cpx a = xp[q], b = xp[q + s * m], c = xp[q + 2 * s * m];
                const cpx t = {b.re + c.re, b.im + c.im}, u = {b.re - c.re, b.im - c.im};
                const cpx e = {a.re - h * t.re, a.im - h * t.im};
                /* -j c3 u */
                const cpx v = {c3 * u.im, -c3 * u.re};
                cpx o1 = {e.re + v.re, e.im + v.im}, o2 = {e.re - v.re, e.im - v.im};
                yp[q].re = a.re + t.re;
                yp[q].im = a.im + t.im;
                yp[q + s] = cmulc(o1, w[0], inv);
                yp[q + 2 * s] = cmulc(o2, w[1], inv);
            }
        } else if (r == 5) {
            const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
            const double s1 = 0.95105651629515357212 * sg, s2 = 0.58778525229247312917 * sg;
            for (long q = 0; q < s; ++q) {
                const cpx a = xp[q], b = xp[q + s * m], c = xp[q + 2 * s * m], d = xp[q + 3 * s * m], e = xp[q + 4 * s * m];
                const cpx t1 = {b.re + e.re, b.im + e.im}, t2 = {c.re + d.re, c.im + d.im};
                const cpx u1 = {b.re - e.re, b.im - e.im}, u2 = {c.re - d.re, c.im - d.im};
                const cpx m1 = {a.re + c1 * t1.re + c2 * t2.re, a.im + c1 * t1.im + c2 * t2.im};
                const cpx m2 = {a.re + c2 * t1.re + c1 * t2.re, a.im + c2 * t1.im + c1 * t2.im};
                /* -j (s1 u1 + s2 u2), -j (s2 u1 - s1 u2) */
                const cpx v1 = {s1 * u1.im + s2 * u2.im, -(s1 * u1.re + s2 * u2.re)};
                const cpx v2 = {s2 * u1.im - s1 * u2.im, -(s2 * u1.re - s1 * u2.re)};
                cpx o1 = {m1.re + v1.re, m1.im + v1.im}, o4 = {m1.re - v1.re, m1.im - v1.im};
                cpx o2 = {m2.re + v2.re, m2.im + v2.im}, o3 = {m2.re - v2.re, m2.im - v2.im};
                yp[q].re = a.re + t1.re + t2.re;
                yp[q].im = a.im + t1.im + t2.im;
                yp[q + s] = cmulc(o1, w[0], inv);
                yp[q + 2 * s] = cmulc(o2, w[1], inv);
                yp[q + 3 * s] = cmulc(o3, w[2], inv);
                yp[q + 4 * s] = cmulc(o4, w[3], inv);
            }
        } else {
            /* odd prime r = 2 h + 1, direct DFT with the conjugate pairs folded: with S_i = a_i + a_(r-i), D_i = a_i - a_(r-i)
             *   y_k, y_(r-k) = a_0 + sum_i cos(t_ik) S_i  -/+  j sg sum_i sin(t_ik) D_i,   t_ik = 2 pi i k / r
             * -- real coefficients on flat double arrays, blocks of q that stay in cache */
            enum { QB = 128 };
            const cpx *F = P->dft[st]; /* [k][i] = (cos t_ik, -sin t_ik) */
            const int h = (r - 1) / 2;
            double *S = P->scratch[st] + (size_t)omp_tid() * (size_t)(2 * h + 2) * 2 * QB, *D = S + (size_t)h * 2 * QB;
            double *R = D + (size_t)h * 2 * QB, *I = R + 2 * QB;
            for (long q0 = 0; q0 < s; q0 += QB) {
                const int nq = (int)(s - q0 < QB ? s - q0 : QB), n2 = 2 * nq;
                const double *x0 = (const double *)(xp + q0);
                for (int i = 1; i <= h; ++i) {
                    const double *xa = (const double *)(xp + s * m * i + q0), *xb = (const double *)(xp + s * m * (r - i) + q0);
                    double *Si = S + (size_t)(i - 1) * 2 * QB, *Di = D + (size_t)(i - 1) * 2 * QB;
                    for (int t = 0; t < n2; ++t) Si[t] = xa[t] + xb[t], Di[t] = xa[t] - xb[t];
                }
                {
                    double *y0 = (double *)(yp + q0);
                    for (int t = 0; t < n2; ++t) R[t] = x0[t];
                    for (int i = 1; i <= h; ++i) {
                        const double *Si = S + (size_t)(i - 1) * 2 * QB;
                        for (int t = 0; t < n2; ++t) R[t] += Si[t];
                    }
                    for (int t = 0; t < n2; ++t) y0[t] = R[t];
                }
                for (int k = 1; k <= h; ++k) {
                    for (int t = 0; t < n2; ++t) R[t] = x0[t], I[t] = 0.0;
                    for (int i = 1; i <= h; ++i) {
                        const double c = F[k * r + i].re, sn = -F[k * r + i].im; /* cos t_ik, sin t_ik */
                        const double *Si = S + (size_t)(i - 1) * 2 * QB, *Di = D + (size_t)(i - 1) * 2 * QB;
                        for (int t = 0; t < n2; ++t) R[t] += c * Si[t], I[t] += sn * Di[t];
                    }
                    cpx *yk = yp + s * k + q0, *yr = yp + s * (r - k) + q0;
                    for (int q = 0; q < nq; ++q) {
                        /* forward: y_k = R - j I, y_(r-k) = R + j I; inverse: the other way round */
                        cpx a = {R[2 * q] + sg * I[2 * q + 1], R[2 * q + 1] - sg * I[2 * q]};
                        cpx b = {R[2 * q] - sg * I[2 * q + 1], R[2 * q + 1] + sg * I[2 * q]};
                        if (m > 1) a = cmulc(a, w[k - 1], inv), b = cmulc(b, w[r - k - 1], inv);
                        yk[q] = a;
                        yr[q] = b;
                    }
                }
            }
        }
    }
}

/* in: x (destroyed), work: same size; returns the buffer that holds the result */
static cpx *transform(const plan_t *P, cpx *x, cpx *work, int inv) {
    long len = P->n, s = 1;
    cpx *a = x, *b = work;
    for (int st = 0; st < P->nst; ++st) {
        stage(P, st, len, s, a, b, inv);
        len /= P->radix[st];
        s *= P->radix[st];
        cpx *t = a;
        a = b;
        b = t;
    }
    return a;
}

/* ---- exported: plain transform (tests) --------------------------------------------------------------------------- */
/* out = fft(in) (inv = 0) or ifft(in) (inv = 1, with 1/n); interleaved re/im doubles; 0 on success */
int bds_oracle_fft(const double *in, double *out, long n, int inv) {
    plan_t *P = plan_make(n);
    if (!P) return -1;
    cpx *a = (cpx *)malloc(sizeof(cpx) * (size_t)n), *b = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    if (!a || !b) {
        free(a);
        free(b);
        plan_free(P);
        return -2;
    }
    memcpy(a, in, sizeof(cpx) * (size_t)n);
    const cpx *r = transform(P, a, b, inv);
    const double sc = inv ? 1.0 / (double)n : 1.0;
    for (long i = 0; i < n; ++i) out[2 * i] = r[i].re * sc, out[2 * i + 1] = r[i].im * sc;
    free(a);
    free(b);
    plan_free(P);
    return 0;
}

/* ---- exported: the Doppler rows of one PRN ---------------------------------------------------------------------------
 * sig_re / sig_im : the first n samples of longSignal as doubles (sig_im NULL for a real record)
 * code_d / code_p : the sampled code tables, x_len samples each (the reference zero-pads them to n: B1C :176-187,
 *                   B2a :179-183); code_p NULL = data component only (B1C with pilotACQflag = 0)
 * frq[nb]         : the Doppler bins to evaluate (frqBins(b), B1C :194-195 / B2a :190-191)
 * kind            : 0 = B2a sum  abs(d) + abs(p)                                   (B2a :204-209)
 *                   1 = B1C      (abs(d) sqrt(11) + abs(p) sqrt(29)) / sqrt(40)    (B1C :216-219); abs(d) alone without pilot (:209-212)
 * row_max / row_arg[nb] : maximum of every row and its first (0-based) index       (max(results, [], 2))
 * col_max[n]      : running maximum over the rows evaluated, NULL to skip          (max(results))   -- updated, not reset
 * rows            : NULL, or nb x n doubles receiving the rows themselves
 * nthreads        : OpenMP threads over the bins (<= 0: the runtime's default)
 * returns 0, or < 0 (-1 length not factorable within the plan, -2 out of memory) */
int bds_oracle_coarse_rows(const double *sig_re, const double *sig_im, long n, double fs, const double *code_d, const double *code_p,
                           long x_len, const double *frq, int nb, int kind, double *row_max, long *row_arg, double *col_max,
                           double *rows, int nthreads) {
    if (n < 1 || x_len < 0 || x_len > n || nb < 0) return -3;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads); /* before the plan: its scratch is per thread */
#else
    (void)nthreads;
#endif
    plan_t *P = plan_make(n);
    if (!P) return -1;
    int rc = 0;
    const int ncomp = code_p ? 2 : 1;
    /* conj(fft(code zero-padded to n)) */
    cpx *cs[2] = {NULL, NULL};
    {
        cpx *w = (cpx *)malloc(sizeof(cpx) * (size_t)n);
        for (int c = 0; c < ncomp && w; ++c) {
            cpx *a = (cpx *)malloc(sizeof(cpx) * (size_t)n);
            if (!a) {
                rc = -2;
                break;
            }
            const double *code = c ? code_p : code_d;
            for (long i = 0; i < n; ++i) a[i].re = i < x_len ? code[i] : 0.0, a[i].im = 0.0;
            cpx *r = transform(P, a, w, 0);
            cs[c] = (cpx *)malloc(sizeof(cpx) * (size_t)n);
            if (!cs[c]) {
                free(a);
                rc = -2;
                break;
            }
            for (long i = 0; i < n; ++i) cs[c][i].re = r[i].re, cs[c][i].im = -r[i].im;
            free(a);
        }
        if (!w) rc = -2;
        free(w);
    }
    const double ts = 1.0 / fs;
    const double w11 = sqrt(11.0), w29 = sqrt(29.0), w40 = sqrt(40.0);
    const double inv_n = 1.0 / (double)n;
    if (rc == 0) {
#pragma omp parallel
        {
            cpx *x = (cpx *)malloc(sizeof(cpx) * (size_t)n), *w = (cpx *)malloc(sizeof(cpx) * (size_t)n);
            cpx *y = (cpx *)malloc(sizeof(cpx) * (size_t)n);
            double *row = (double *)malloc(sizeof(double) * (size_t)n);
            double *cm = col_max ? (double *)malloc(sizeof(double) * (size_t)n) : NULL;
            int bad = !x || !w || !y || !row || (col_max && !cm);
            if (bad) {
#pragma omp atomic write
                rc = -2;
            }
            if (cm)
                for (long i = 0; i < n; ++i) cm[i] = -INFINITY;
#pragma omp for schedule(dynamic, 1)
            for (int b = 0; b < nb; ++b) {
                if (bad) continue;
                /* carr = exp(1i * frqBins(b) * phasePoints), phasePoints = (0 : n-1) * 2 * pi * ts   (B1C :144,198; B2a :146,194) */
                const double f = frq[b];
                for (long k = 0; k < n; ++k) {
                    const double pp = (((double)k * 2.0) * M_PI) * ts;
                    const double a = f * pp;
                    const double c = cos(a), sn = sin(a);
                    const double re = sig_re[k], im = sig_im ? sig_im[k] : 0.0;
                    x[k].re = c * re - sn * im;
                    x[k].im = c * im + sn * re;
                }
                const cpx *X = transform(P, x, w, 0); /* :201-205 */
                cpx *other = X == x ? w : x;
                for (int c = 0; c < ncomp; ++c) {
                    for (long i = 0; i < n; ++i) y[i] = cmul(X[i], cs[c][i]);
                    const cpx *r = transform(P, y, other, 1); /* ifft :209-218 */
                    if (c == 0) {
                        for (long i = 0; i < n; ++i) row[i] = hypot(r[i].re * inv_n, r[i].im * inv_n);
                        if (kind == 1 && ncomp == 2)
                            for (long i = 0; i < n; ++i) row[i] = row[i] * w11;
                    } else if (kind == 1) {
                        for (long i = 0; i < n; ++i) row[i] = (row[i] + hypot(r[i].re * inv_n, r[i].im * inv_n) * w29) / w40;
                    } else {
                        for (long i = 0; i < n; ++i) row[i] = row[i] + hypot(r[i].re * inv_n, r[i].im * inv_n);
                    }
                    /* (the inverse transforms ping-pong between y and `other`; X stays intact for the second component) */
                }
                double mx = -INFINITY;
                long arg = 0;
                for (long i = 0; i < n; ++i)
                    if (row[i] > mx) mx = row[i], arg = i;
                row_max[b] = mx;
                row_arg[b] = arg;
                if (cm)
                    for (long i = 0; i < n; ++i)
                        if (row[i] > cm[i]) cm[i] = row[i];
                if (rows) memcpy(rows + (size_t)b * (size_t)n, row, sizeof(double) * (size_t)n);
            }
            if (cm && !bad) {
#pragma omp critical
                for (long i = 0; i < n; ++i)
                    if (cm[i] > col_max[i]) col_max[i] = cm[i];
            }
            free(x);
            free(w);
            free(y);
            free(row);
            free(cm);
        }
    }
    free(cs[0]);
    free(cs[1]);
    plan_free(P);
    return rc;
}

int bds_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
