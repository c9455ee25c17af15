"""NumPy model of the wave-private inverse row pass for 4096-point rows (csrc/bds_acq_wrows.h): stage decomposition, lane maps,
LDS layouts and bank conflicts.  Decimation in frequency, so that the LAST stage leaves a thread 16 outputs e'' + 256 p' with
e'' = thread id (coalesced stores); the scatter is on the loads instead (16-byte pieces).

  r = 16 b' + q',  b' = bl + 16 bh;   e = e'' + 256 p',  e'' = u + 16 v
  Y_q'[u + 16 v] = sum_bl w16^(bl v) w256^(bl u) sum_bh w16^(bh u) x[16 (bl + 16 bh) + q']        (wave-private: wave w has q' = 4w..4w+3)
  X[e'' + 256 p'] = sum_q' w16^(q' p') w4096^(q' e'') Y_q'[e'']                                     (cooperative, thread e'')
  phase 1a: lane (ql = lane & 3, bl = lane >> 2): radix 16 over bh, to the wave's region at [64 u + lane]
  phase 1b: lane (ql, u = lane >> 2): reads row u starting at column u (rotation: factor w16^(-u v), folded into the inter-pass
            twiddle), twiddle w256^(bl u) on these inputs, radix 16 over bl, Y to the exchange buffer at [20 e'' + ((e'' >> 3) & 3) + q']
  phase 2 : thread e'': radix 16 over q' with twiddle w4096^(q' e'')
"""
import numpy as np

S = 4096
rng = np.random.default_rng(0)
x = rng.standard_normal(S) + 1j * rng.standard_normal(S)
ref = np.fft.ifft(x) * S
w = lambda n, k: np.exp(2j * np.pi * k / n)
dft16 = np.array([[w(16, q * p) for q in range(16)] for p in range(16)])
reg = np.zeros((4, 1024), complex)
ex = np.zeros(256 * 20 + 4, complex)
conf = {}


def bank(name, addrs, group):
    # MI355X_MICROARCH.md, LDS: a ds_read_b64 is served in two halves of 32 lanes over 64 banks (32 eight-byte slots), a
    # ds_write_b64 in four groups of 16 contiguous lanes over 32 banks (16 slots)
    assert group == (16 if "write" in name else 32)
    worst = 1
    for i in range(0, 64, group):
        banks = {}
        for a in addrs[i:i + group]:
            banks.setdefault(a % group, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    conf[name] = max(conf.get(name, 1), worst)


XS = 20
xa = lambda e2: XS * e2 + ((e2 >> 3) & 3)   # exchange buffer: Y_q'[e''] at [20 e'' + ((e'' >> 3) & 3) + q']


for wave in range(4):
    for lane in range(64):
        ql, bl = lane & 3, lane >> 2
        q = 4 * wave + ql
        A = dft16 @ np.array([x[16 * (bl + 16 * bh) + q] for bh in range(16)])
        for u in range(16):
            reg[wave, 64 * u + lane] = A[u]
    for u in range(16):
        bank("1a.write", [64 * u + l for l in range(64)], 16)
    wr = {}
    for lane in range(64):
        ql, u = lane & 3, lane >> 2
        q = 4 * wave + ql
        # the twiddle w256^(bl u) is applied to the INPUTS of this butterfly (bds_fft_fma.h folds it into the first layer), less
        # the common factor w256^(u u) so that input 0 needs none; the factor goes into the inter-pass twiddle
        a = np.array([reg[wave, 64 * u + 4 * ((j + u) & 15) + ql] * w(256, u * (((j + u) & 15) - u)) for j in range(16)])
        Yr = dft16 @ a
        for v in range(16):
            ad = xa(u + 16 * v) + q
            ex[ad] = Yr[v]   # = w16^(-u v) w256^(-u u) Y_q[u + 16 v]
            wr.setdefault(v, []).append(ad)
    for j in range(16):
        bank("1b.read", [64 * (l >> 2) + 4 * ((j + (l >> 2)) & 15) + (l & 3) for l in range(64)], 32)
    for v, ad in wr.items():
        bank("1b.write", ad, 16)
out = np.zeros(S, complex)
for e2 in range(256):
    u, v = e2 & 15, e2 >> 4
    z = np.array([ex[xa(e2) + q] * w(4096, q * e2) for q in range(16)])
    X = dft16 @ z
    for p in range(16):
        out[e2 + 256 * p] = X[p] * w(16, u * v) * w(256, u * u)   # rotation and left-out factors, folded into the inter-pass twiddle
for q in range(16):
    for wv in range(4):
        bank("2.read", [xa(64 * wv + l) + q for l in range(64)], 32)
err = np.abs(out - ref).max() / np.abs(ref).max()
print(f"max rel err {err:.2e}; LDS {(4 * 1024 + 256 * XS + 4) * 8} B per workgroup; worst bank conflict per access class: {conf}")
assert err < 1e-12
