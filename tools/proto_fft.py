"""NumPy prototype of the index maps used by csrc/bds_fft.hip (development aid).

Checks (a) the mixed-radix Stockham autosort stage recurrence, (b) the two-pass
(4-step) forward/inverse split with the [k1][k2] spectrum layout and (c) that the
zero-padded length-L linear correlation reproduces the reference's circular
length-N correlation lag for lag.
"""
import numpy as np


def stockham(x, radices, sign):
    S = x.size
    a = x.astype(np.complex128).copy()
    Ns = 1
    for R in radices:
        nb = S // R
        out = np.empty_like(a)
        for j in range(nb):
            k = j % Ns
            v = np.array([a[j + q * nb] * np.exp(sign * 2j * np.pi * q * k / (Ns * R)) for q in range(R)])
            V = np.array([sum(v[q] * np.exp(sign * 2j * np.pi * q * p / R) for q in range(R)) for p in range(R)])
            j0 = (j // Ns) * Ns * R + k
            for p in range(R):
                out[j0 + p * Ns] = V[p]
        a = out
        Ns *= R
    return a


def fwd2(x, L1, L2):
    """x natural [n = n1*L2 + n2] -> spectrum stored [k1][k2] holding X[k1 + L1*k2]."""
    L = L1 * L2
    a = x.reshape(L1, L2)
    A = np.fft.fft(a, axis=0)                      # over n1 -> k1
    k1 = np.arange(L1)[:, None]; n2 = np.arange(L2)[None, :]
    A = A * np.exp(-2j * np.pi * k1 * n2 / L)
    return np.fft.fft(A, axis=1)                   # over n2 -> k2 ; [k1][k2]


def inv2(Z, L1, L2):
    """Z stored [k1][k2] -> y natural [n1*L2 + n2] (unnormalised inverse)."""
    L = L1 * L2
    B = np.fft.ifft(Z, axis=1) * L2                # over k2 -> n2
    k1 = np.arange(L1)[:, None]; n2 = np.arange(L2)[None, :]
    B = B * np.exp(+2j * np.pi * k1 * n2 / L)
    y = np.fft.ifft(B, axis=0) * L1                # over k1 -> n1
    return y.reshape(-1)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for rad in ([2, 3, 4, 5], [5, 5, 4], [4, 4, 2, 3], [3, 5, 2]):
        S = int(np.prod(rad))
        x = rng.normal(size=S) + 1j * rng.normal(size=S)
        for sign in (-1, +1):
            ref = np.fft.fft(x) if sign < 0 else np.fft.ifft(x) * S
            assert np.allclose(stockham(x, rad, sign), ref), (rad, sign)
    L1, L2 = 12, 20
    x = rng.normal(size=L1 * L2) + 1j * rng.normal(size=L1 * L2)
    X = fwd2(x, L1, L2)
    Xn = np.fft.fft(x)
    k1 = np.arange(L1)[:, None]; k2 = np.arange(L2)[None, :]
    assert np.allclose(X, Xn[k1 + L1 * k2])
    assert np.allclose(inv2(X, L1, L2), x * L1 * L2)
    # linear-correlation equivalence
    N, Xl = 100, 50
    L1, L2 = 10, 15
    L = L1 * L2
    assert L >= N + Xl - 1
    y = rng.normal(size=N) + 1j * rng.normal(size=N)
    c = np.concatenate([rng.choice([-1.0, 1.0], Xl), np.zeros(N - Xl)])
    ref = np.fft.ifft(np.fft.fft(y) * np.conj(np.fft.fft(c)))
    yext = np.zeros(L, complex); n = np.arange(N + Xl - 1); yext[n] = y[n % N]
    cext = np.zeros(L); cext[:Xl] = c[:Xl]
    r = inv2(fwd2(yext, L1, L2) * np.conj(fwd2(cext, L1, L2)), L1, L2) / L
    assert np.allclose(r[:N], ref)
    print("proto ok")
