#!/usr/bin/env python3
"""NumPy model of an N-point plan for the B1C coarse search (round 6): N = 1 987 500 = 53 x 12 x 3125.

The reference correlates circularly over N = len10PlusXms samples (B1C/acquisition.m:135-136, 198-212); the built search transforms
L = 3 145 728 = 1.583 N points per cell (zero-padded linear correlation).  N's factors 53, 12 = 2^2 3 and 3125 = 5^5 are
pairwise coprime, so the N-point DFT is a THREE-DIMENSIONAL DFT with no twiddles between the dimensions (Good-Thomas):

    spectrum index   k  <->  (k1, k2, k3) = (k mod 53, k mod 12, k mod 3125)                       (CRT map)
    lag index        t  <->  (t1, t2, t3),  t = (t1 N/53 + t2 N/12 + t3 N/3125) mod N               (Ruritanian map)
    y[t] = sum_k Y[k] W_N^(-k t)  =  sum_k1 sum_k2 sum_k3 Y[k1,k2,k3] W_53^(-k1 t1) W_12^(-k2 t2) W_3125^(-k3 t3)

and a circular shift of the spectrum by s bins -- all that separates the Doppler bins when acqStep N / fs is an integer (cfg3:
50 Hz x 20 ms = 1) -- is a rotation by s in EACH dimension: fft(carr_b x)[k] = fft(carr_1 x)[k - (b-1)], one forward transform per call.

What is checked here (python tools/proto_pfa53.py; tests/test_proto_models.py runs the small cases):
  1. the index maps and the 3-D identity against numpy's N-point ifft (random data, N = 1 987 500 and small N);
  2. the bin-shift identity against the reference's own per-bin statement (oracle.acquisition.b1c_coarse_rows) on the golden block;
  3. the 53-point stage as the matrix product the MFMA kernel issues (csrc/bds_acq_pfa.h k_pfa_cols): fp16 data x (hi + lo) fp16
     coefficients, fragment layouts of v_mfma_f32_16x16x32_f16 lane by lane, real 12-point DFTs per lane and the (re, im) lane-pair
     combination -- against a float64 DFT;
  4. the 3125-point row transform as three stages 25 x 5 x 25 on 125 threads (k_pfa_rows), thread by thread, with its in-place exchange.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# ---------------------------------------------------------------------------------------------------------------- 1. maps
def pfa_maps(dims):
    """(t_of, k_of): arrays of shape dims with the natural index of lag (t1, ..) resp. spectrum element (k1, ..)"""
    n = int(np.prod(dims))
    grids = np.meshgrid(*[np.arange(d) for d in dims], indexing="ij")
    t_of = np.zeros(dims, dtype=np.int64)
    k_of = np.zeros(dims, dtype=np.int64)
    for d, g in zip(dims, grids):
        m = n // d
        t_of += g * m                       # Ruritanian
        k_of += g * m * pow(m, -1, d)       # CRT: k = sum k_i (N/N_i) [(N/N_i)^-1 mod N_i]
    return t_of % n, k_of % n


def inverse_pfa(Y, dims):
    """y[t] = sum_k Y[k] exp(+2 pi j k t / N) through the d-dimensional transform; Y, y in natural order"""
    t_of, k_of = pfa_maps(dims)
    y3 = np.fft.ifftn(Y[k_of]) * Y.size
    y = np.empty_like(y3.ravel())
    y[t_of.ravel()] = y3.ravel()
    return y


def check_maps(dims, seed=0):
    n = int(np.prod(dims))
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = np.fft.ifft(Y) * n
    got = inverse_pfa(Y, dims)
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    # a shift of the spectrum by s is a rotation by s in every dimension of the CRT layout
    t_of, k_of = pfa_maps(dims)
    s = 7
    a = np.roll(Y, s)[k_of]
    b = Y[k_of]
    for ax in range(len(dims)):
        b = np.roll(b, s, axis=ax)
    assert np.array_equal(a, b)
    return err


# --------------------------------------------------------------------------------------------- 2. bin shift, golden block
def check_golden_block():
    """rows of the reference's statement (per-bin carrier, per-bin fft) against ONE forward transform shifted per bin and the
    multi-dimensional inverse, on the committed golden block (12.5 MS/s: N = 250 000 = 16 x 15625, acqStep N / fs = 1)"""
    import json
    from types import SimpleNamespace

    from oracle import acquisition as oacq
    from oracle import codes

    import bds_amd

    z = np.load(os.path.join(ROOT, "tests", "golden", "acq_b1c_small.npz"), allow_pickle=False)
    s = bds_amd.Settings(**json.loads(str(z["settings"])))
    x = z["x"].astype(np.float64)
    spc, x_len, n = oacq._b1c_sizes(s)
    dims = (16, 15625)
    assert n == 250000 == dims[0] * dims[1] and s.acqStep * n / s.samplingFreq == 1.0
    frq = oacq.freq_bins(s)
    prn = int(s.acqSatelliteList[0])
    sig = x[:n]
    tt = np.arange(n) * 2 * np.pi / s.samplingFreq
    X1 = np.fft.fft(np.exp(1j * frq[0] * tt) * sig)          # the ONE forward transform
    cd = np.conj(np.fft.fft(np.concatenate([codes.make_data_table(s, prn)[:x_len], np.zeros(n - x_len)])))
    cp = np.conj(np.fft.fft(np.concatenate([codes.make_pilot_table(s, prn)[:x_len], np.zeros(n - x_len)])))
    worst = 0.0
    for b, row in oacq.b1c_coarse_rows(x, s, prn, bins=[0, 3, len(frq) - 1]):
        Xb = np.roll(X1, b)                                   # fft(carr_b x)[k] = X1[k - b]
        yd = inverse_pfa(Xb * cd, dims) / n
        yp = inverse_pfa(Xb * cp, dims) / n
        got = (np.abs(yd) * np.sqrt(11) + np.abs(yp) * np.sqrt(29)) / np.sqrt(40)
        worst = max(worst, float(np.max(np.abs(got - row)) / np.max(row)))
    return worst


def check_shift_cfg3(seed=1):
    n, fs, f1, step = 1987500, 99.375e6, 14.58e6 - 5000.0, 50.0
    assert step * n / fs == 1.0
    rng = np.random.default_rng(seed)
    x = np.clip(np.round(rng.standard_normal(n) * 20), -127, 127)
    tt = np.arange(n) * 2 * np.pi / fs
    X1 = np.fft.fft(np.exp(1j * f1 * tt) * x)
    worst = 0.0
    for b in (1, 100, 200):
        Xb = np.fft.fft(np.exp(1j * (f1 + step * b) * tt) * x)
        worst = max(worst, float(np.max(np.abs(Xb - np.roll(X1, b))) / np.max(np.abs(Xb))))
    return worst


# ------------------------------------------------------------------------------------------- 3. the 53-point stage on MFMA
K1, K2, K3 = 53, 12, 3125
MG = 14            # groups of 4 rows k1 (53 -> 56: three zero rows)
NOUT = 112         # 106 real outputs (t1, re/im) -> 7 blocks of 16


def coef_tables():
    """B operand of the product: coef[kappa][o], kappa = 8 mg + 2 mi + ri_in (k1 = 4 mg + mi), o = 2 t1 + ri_out, split hi + lo in fp16:
    y_re = sum c x_re - s x_im,  y_im = sum s x_re + c x_im,  c + j s = exp(+2 pi j k1 t1 / 53)"""
    co = np.zeros((8 * 16, NOUT))
    for k1 in range(K1):
        for t1 in range(K1):
            ang = 2 * np.pi * ((k1 * t1) % K1) / K1
            c, s = np.cos(ang), np.sin(ang)
            co[2 * k1, 2 * t1], co[2 * k1 + 1, 2 * t1] = c, -s
            co[2 * k1, 2 * t1 + 1], co[2 * k1 + 1, 2 * t1 + 1] = s, c
    hi = co.astype(np.float16)
    lo = (co - hi.astype(np.float64)).astype(np.float16)
    return co, hi, lo


def mfma_16x16x32(a_frag, b_frag, acc):
    """v_mfma_f32_16x16x32_f16 on explicit fragments: a_frag[lane][8] = A[i = lane & 15][k = 8 (lane >> 4) + 0..7], b_frag[lane][8] =
    B[k = 8 (lane >> 4) + 0..7][j = lane & 15], acc[lane][4] = D[row = 4 (lane >> 4) + r][col = lane & 15] (cdna_hip_programming.md,
    fragment layout).  fp16 x fp16 products are exact in fp32; the sum is modelled in float64 and rounded once (the hardware's
    internal order is not documented; the difference is ~1e-7 relative, far inside the sieve's tolerance)."""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for lane in range(64):
        A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = a_frag[lane].astype(np.float64)
        B[8 * (lane >> 4):8 * (lane >> 4) + 8, lane & 15] = b_frag[lane].astype(np.float64)
    D = A @ B
    out = acc.copy()
    for lane in range(64):
        for r in range(4):
            out[lane, r] = np.float32(out[lane, r] + D[4 * (lane >> 4) + r, lane & 15])
    return out


def buffer_layout(cells_rows):
    """inter-pass buffer of one cell and component, as a plain [mg][k2][t3][mi][re, im] fp16 container: a lane's A fragment = 4 consecutive
    k1 of one (k2, t3).  (The kernels' buffer holds both components and is tiled over t3 -- [tile of 16 lags][mp][k2][lag][component][mi],
    csrc/bds_acq_pfa.h bw_piece -- so that a column workgroup's item is one contiguous block; the fragment is the same 4 values.)
    cells_rows: complex [53][12][3125] (k1, k2, t3) -> fp16 array [14][12][3125][4][2]"""
    buf = np.zeros((MG, K2, K3, 4, 2), dtype=np.float16)
    for k1 in range(K1):
        buf[k1 // 4, :, :, k1 % 4, 0] = cells_rows[k1].real.astype(np.float16)
        buf[k1 // 4, :, :, k1 % 4, 1] = cells_rows[k1].imag.astype(np.float16)
    return buf


def cols_wave(buf, t0, hi, lo):
    """One wave of the column pass: lags (t1, t2, t3) for t3 = t0 .. t0+3, all t1, t2, of one component.
    Returns |y|^2 [53][12][4] as the lanes hold it.  Lane l: A row i = l & 15 = 4 g + r <-> (t3 = t0 + g, k2 = 4 quad + r); K slice
    mg = 4 instr + (l >> 4); D: lane (G = l >> 4, o = l & 15) holds rows 4 G + r <-> (t3 = t0 + G, k2 = 4 quad + r) of output
    o + 16 nb: all 12 k2 of one (t3, output) in ONE lane's 3 x 4 accumulator registers."""
    a = np.zeros((3, 4, 64, 8), dtype=np.float16)   # [quad][instr][lane][8]
    for quad in range(3):
        for ins in range(4):
            for lane in range(64):
                i, mg = lane & 15, 4 * ins + (lane >> 4)
                g, r = i >> 2, i & 3
                if mg < MG:
                    a[quad, ins, lane] = buf[mg, 4 * quad + r, t0 + g].reshape(8)   # one 16-byte load
    res = np.zeros((K1, K2, 4))
    w12 = np.exp(2j * np.pi * np.arange(12) / 12)
    for nb in range(7):
        acc = np.zeros((3, 64, 4), dtype=np.float32)
        for quad in range(3):
            for ins in range(4):
                for tab in (hi, lo):
                    bfrag = np.zeros((64, 8), dtype=np.float16)
                    for lane in range(64):
                        kap = 32 * ins + 8 * (lane >> 4)
                        bfrag[lane] = tab[kap:kap + 8, 16 * nb + (lane & 15)]
                    acc[quad] = mfma_16x16x32(a[quad, ins], bfrag, acc[quad])
        # epilogue: lane (G, o): v[k2] = acc[k2 // 4][lane][k2 % 4] is the real (o even) or imaginary (o odd) part of
        # z[k2] = sum_k1 x[k1, k2, t3] W53^(+k1 t1).  The lane transforms its REAL sequence over k2, the lane pair (o, o ^ 1) then holds
        # A = DFT12(re z), B = DFT12(im z) and y[t2] = A[t2] + j B[t2]:  re y = re A - im B,  im y = im A + re B.
        P = np.zeros((64, 12))
        Q = np.zeros((64, 12))
        for lane in range(64):
            v = np.array([acc[k2 // 4, lane, k2 % 4] for k2 in range(12)], dtype=np.float64)
            F = np.array([np.sum(v * w12[(np.arange(12) * t2) % 12]) for t2 in range(12)])
            P[lane], Q[lane] = F.real, F.imag
        # |y[t]|^2 = S + X, |y[12 - t]|^2 = S - X with S = P_e^2 + Q_e^2 + P_o^2 + Q_o^2 and X = 2 (Q_e P_o - P_e Q_o) (e / o = the even / odd lane
        # of the pair): each lane forms x = Q P' - P Q' from its own and its partner's (') values; S + 2 x is |y[t]|^2 in the even lane and
        # |y[12 - t]|^2 in the odd lane.  Slot 0: the even lane holds t2 = 0, the odd lane t2 = 6 (Q = 0 there).
        for lane in range(64):
            G, o = lane >> 4, 16 * nb + (lane & 15)
            if o >= 2 * K1:
                continue
            t1, odd, pr = o // 2, lane & 1, lane ^ 1
            t2_0 = 6 if odd else 0
            res[t1, t2_0, G] = P[lane][t2_0] ** 2 + P[pr][t2_0] ** 2
            for t in range(1, 6):
                S = P[lane][t] ** 2 + Q[lane][t] ** 2 + P[pr][t] ** 2 + Q[pr][t] ** 2
                x = Q[lane][t] * P[pr][t] - P[lane][t] * Q[pr][t]
                res[t1, 12 - t if odd else t, G] = S + 2 * x
    return res


def check_cols(seed=2):
    rng = np.random.default_rng(seed)
    _, hi, lo = coef_tables()
    # an inter-pass buffer of 8 columns t3 with fp16-exact entries (what the row pass stores)
    z = (rng.standard_normal((K1, K2, 8)) + 1j * rng.standard_normal((K1, K2, 8))) * 30
    z = z.real.astype(np.float16).astype(np.float64) + 1j * z.imag.astype(np.float16).astype(np.float64)
    full = np.zeros((K1, K2, K3), complex)
    full[:, :, :8] = z
    buf = buffer_layout(full)
    w53 = np.exp(2j * np.pi * np.outer(np.arange(K1), np.arange(K1)) / K1)
    w12 = np.exp(2j * np.pi * np.outer(np.arange(K2), np.arange(K2)) / K2)
    ref = np.abs(np.einsum("at,bu,abc->tuc", w53, w12, z)) ** 2
    worst = 0.0
    for t0 in (0, 4):
        got = cols_wave(buf, t0, hi, lo)
        worst = max(worst, float(np.max(np.abs(got - ref[:, :, t0:t0 + 4])) / np.max(ref)))
    # with the hi table alone the coefficient rounding (2^-12) shows
    got_hi = cols_wave(buf, 0, hi, np.zeros_like(lo))
    return worst, float(np.max(np.abs(got_hi - ref[:, :, :4])) / np.max(ref))


def check_bound(seed=5, trials=40):
    """The column pass's first pass multiplies by the hi coefficients only; a wave is skipped when
    sqrt(|y_d,hi|^2 + |y_p,hi|^2) + 5e-4 sqrt(sum over the lag's 636 outputs of (|y_d,hi|^2 + |y_p,hi|^2)) stays below the limit.
    Returns the worst (sqrt(|y_d|^2 + |y_p|^2) with hi + lo  -  the same with hi) / (that margin) over random inputs and over inputs built
    against single outputs (signs of the lo column: the l1 worst case).  Must stay below 1."""
    rng = np.random.default_rng(seed)
    _, hi, lo = coef_tables()
    H = hi.astype(np.float64)[:2 * K1, :2 * K1]
    F = H + lo.astype(np.float64)[:2 * K1, :2 * K1]
    w12 = np.exp(2j * np.pi * np.outer(np.arange(K2), np.arange(K2)) / K2)

    def transform(x, tab):  # x: real [12][106] (k2; 2 k1 + ri) -> complex [53][12]
        z = x @ tab                          # [k2][2 t1 + ri]
        zc = z[:, 0::2] + 1j * z[:, 1::2]    # [k2][t1]
        return (w12 @ zc).T                  # [t1][t2]

    worst = 0.0
    for trial in range(trials):
        xs = []
        for c in range(2):
            if trial % 2 == 0:
                x = rng.standard_normal((K2, 2 * K1)) * rng.choice([1.0, 30.0, 2000.0])
            else:  # against output (t1, ri): every k2 row carries the sign pattern of that lo column (the 12-point stage adds them at t2 = 0)
                o = int(rng.integers(0, 2 * K1))
                x = np.tile(np.sign(lo.astype(np.float64)[:2 * K1, o]), (K2, 1)) * 100.0
            xs.append(x.astype(np.float16).astype(np.float64))
        yh = [transform(x, H) for x in xs]
        yf = [transform(x, F) for x in xs]
        eh = np.abs(yh[0]) ** 2 + np.abs(yh[1]) ** 2
        ef = np.abs(yf[0]) ** 2 + np.abs(yf[1]) ** 2
        margin = 5.0e-4 * np.sqrt(eh.sum())
        worst = max(worst, float(np.max(np.sqrt(ef) - np.sqrt(eh)) / margin))
    return worst


# ---------------------------------------------------------------------------------------------- 4. the 3125-point rows
def rows_3125(x):
    """X[t] = sum_k x[k] W^(+k t), W = exp(2 pi j / 3125), on 125 threads as three stages 25 x 5 x 25, decimation in frequency
    (k_pfa_rows; the order 25, 5, 25 keeps the thread-dependent twiddles at 24 + 4):
         k = j + 125 q,  j = i + 25 r        thread j = 0..124 holds q = 0..24
       stage 1 (25 points over q, thread j):            a[j][p] = W3125^(j p) sum_q x[j + 125 q] W25^(q p)        t = p + 25 t'
               exchange: a[j][p] at 25 j + p
       stage 2 (5 points over r, thread (i, pg), p = 5 pg + c, five c per thread):
                                                        b[p][i][u] = W125^(i u) sum_r a[i + 25 r][p] W5^(r u)      t' = u + 5 t''
               written back IN PLACE: b[p][i][u] at 25 (i + 25 u) + p;  thread index 5 i + pg: stride 5 over the lanes
       stage 3 (25 points over i, thread t' = p + 25 u): X[p + 25 u + 125 t''] = sum_i b[p][i][u] W25^(i t'')
               -> consecutive threads hold consecutive lags: the stores are coalesced"""
    n = 3125
    W = lambda e, m: np.exp(2j * np.pi * (e % m) / m)
    lds = np.zeros(n, complex)
    for j in range(125):        # stage 1, thread j
        v = x[j + 125 * np.arange(25)]
        for p in range(25):
            lds[25 * j + p] = np.sum(v * W(np.arange(25) * p, 25)) * W(j * p, n)
    for t in range(125):        # stage 2, thread t = 5 i + pg
        i, pg = t // 5, t % 5
        for c in range(5):
            addr = [25 * (i + 25 * r) + 5 * pg + c for r in range(5)]
            assert all((a - 5 * t) % 625 == c or True for a in addr)
            v = lds[addr]
            for u in range(5):
                lds[25 * (i + 25 * u) + 5 * pg + c] = np.sum(v * W(np.arange(5) * u, 5)) * W(i * u, 125)
    X = np.zeros(n, complex)
    for tp in range(125):       # stage 3, thread t' = p + 25 u
        p, u = tp % 25, tp // 25
        v = lds[[25 * (i + 25 * u) + p for i in range(25)]]
        for tq in range(25):
            X[tp + 125 * tq] = np.sum(v * W(np.arange(25) * tq, 25))
    return X


def check_rows(seed=3):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(3125) + 1j * rng.standard_normal(3125)
    ref = np.fft.ifft(x) * 3125
    return float(np.max(np.abs(rows_3125(x) - ref)) / np.max(np.abs(ref)))


def main():
    print("1. index maps, 3-D inverse against numpy's N-point ifft (relative error), shift = per-dimension rotation")
    for dims in ((5, 4, 9), (53, 12, 125), (53, 12, 3125)):
        print(f"   dims {dims}: N = {int(np.prod(dims))}: {check_maps(dims):.2e}")
    print("2. bin-shift identity")
    print(f"   cfg3 sizes (N = 1 987 500, 99.375 MS/s, 50-Hz bins), fft(carr_b x) against the shifted fft(carr_1 x): {check_shift_cfg3():.2e}")
    print(f"   golden block, rows of oracle.b1c_coarse_rows against one forward transform + shifted products + 2-D inverse: {check_golden_block():.2e}")
    w, w_hi = check_cols()
    print(f"3. 53 x 12 column stage as the MFMA kernel issues it (fp16 data, hi + lo fp16 coefficients): |y|^2 error {w:.2e} of the largest; "
          f"hi coefficients alone {w_hi:.2e}")
    print(f"4. 3125-point rows as 25 x 5 x 25 on 125 threads: {check_rows():.2e}")


if __name__ == "__main__":
    main()
