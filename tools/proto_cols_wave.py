"""NumPy model of the wave-private inverse column pass (csrc/bds_acq_wcols.h): checks the stage
decomposition, the lane mappings and the LDS layouts (incl. bank-conflict counts) lane by lane.

S = 64 R1 points per column, tile of 8 columns on 4 waves:
  phase A (cooperative): wave i, lane (cp = lane & 3, bq = lane >> 2) takes butterfly b = 16 i + bq of column
      pair cp: radix-R1 over q of x[b + 64 q], twiddle w_S^(b p), written to region R_cp at [m = c R1 + p][b]
  phase B (wave w owns region R_w, columns 2w, 2w+1): lane (ml = lane & 7, bl = lane >> 3), slots s: m = ml + 8 s
      stage 2: radix-8 over bh of z[m][bl + 8 bh], written back IN PLACE at [m][8 u + bl]
      stage 3: lane (ml, u = lane >> 3): twiddle w_64^(bl u) on the inputs, radix-8 over bl -> X[p + R1 (u + 8 v)]; the lane
      reads its row starting at column u (conflict-free): a rotation of the inputs, i.e. a unit factor on the outputs,
      invisible in |X| (like the factor w_64^(u u) the twiddles leave out)
"""
import sys

import numpy as np


def run(S, seed=0, verbose=True):
    R1 = S // 64
    M = 2 * R1
    SL = M // 8
    MS = 68                      # elements between consecutive m inside a region
    RS = M * MS + 4              # region stride in elements (4 mod 16 eight-byte slots: phase A's writers)
    sw = lambda m: (m & 7) >> 1  # element (m, column) lives at [m MS + (column ^ sw(m))]
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((S, 8)) + 1j * rng.standard_normal((S, 8))
    ref = np.fft.ifft(x, axis=0) * S   # X[e] = sum_r x[r] exp(+2 pi j r e / S)
    wS = lambda k: np.exp(2j * np.pi * (k % S) / S)
    lds = np.zeros(4 * RS, complex)
    conflicts = {}

    def bank_check(name, addrs):
        # addrs: 64 element addresses (8-byte elements).  MI355X_MICROARCH.md, LDS: a ds_read_b64 is served in two halves of 32
        # lanes over 64 banks (32 eight-byte slots), a ds_write_b64 in four groups of 16 contiguous lanes over 32 banks (16 slots)
        group, slots = (16, 16) if "write" in name else (32, 32)
        worst = 1
        for i in range(0, 64, group):
            banks = {}
            for a in addrs[i:i + group]:
                banks.setdefault(a % slots, set()).add(a)
            worst = max(worst, max(len(v) for v in banks.values()))
        conflicts[name] = max(conflicts.get(name, 1), worst)

    # ---- phase A
    for i in range(4):
        wr = {}
        for lane in range(64):
            cp, bq = lane & 3, lane >> 2
            b = 16 * i + bq
            for c in range(2):
                col = 2 * cp + c
                v = np.array([x[b + 64 * q, col] for q in range(R1)])
                Y = np.array([sum(v[q] * np.exp(2j * np.pi * q * p / R1) for q in range(R1)) for p in range(R1)])
                for p in range(R1):
                    m = c * R1 + p
                    a = cp * RS + m * MS + (b ^ sw(m))
                    lds[a] = Y[p] * wS(b * p)
                    wr.setdefault(m, []).append(a)
        for m, addrs in wr.items():
            bank_check("A.write", addrs)
    out = np.zeros((S, 8), complex)
    # ---- phase B
    for w in range(4):
        base = w * RS
        # stage 2: all reads first (the writes are in place)
        regs = {}
        for lane in range(64):
            ml, bl = lane & 7, lane >> 3
            for s in range(SL):
                regs[(lane, s)] = np.array([lds[base + ml * MS + (bl ^ sw(ml)) + s * 8 * MS + 8 * bh] for bh in range(8)])
        for s in range(SL):
            for bh in range(8):
                bank_check("B.read2", [base + (l & 7) * MS + ((l >> 3) ^ sw(l & 7)) + s * 8 * MS + 8 * bh for l in range(64)])
        for lane in range(64):
            ml, bl = lane & 7, lane >> 3
            for s in range(SL):
                z = regs[(lane, s)]
                A = np.array([sum(z[bh] * np.exp(2j * np.pi * bh * u / 8) for bh in range(8)) for u in range(8)])
                for u in range(8):
                    lds[base + ml * MS + s * 8 * MS + 8 * u + (bl ^ sw(ml))] = A[u]
        for s in range(SL):
            for u in range(8):
                bank_check("B.write2", [base + (l & 7) * MS + s * 8 * MS + 8 * u + ((l >> 3) ^ sw(l & 7)) for l in range(64)])
        # stage 3
        for lane in range(64):
            ml, u = lane & 7, lane >> 3
            for s in range(SL):
                m = ml + 8 * s
                c, p = divmod(m, R1)
                # register j holds bl = j; the stage-2 twiddle w_64^(bl u) is applied here, to the inputs (bds_fft_fma.h folds it
                # into the butterfly's first layer)
                a = np.array([lds[base + ml * MS + 8 * u + s * 8 * MS + (j ^ sw(ml))] * wS(R1 * u * j) for j in range(8)])
                X = np.array([sum(a[j] * np.exp(2j * np.pi * j * v / 8) for j in range(8)) for v in range(8)])
                for v in range(8):
                    out[p + R1 * (u + 8 * v), 2 * w + c] = X[v]
        for s in range(SL):
            for bl in range(8):
                bank_check("B.read3", [base + (l & 7) * MS + 8 * (l >> 3) + s * 8 * MS + (bl ^ sw(l & 7)) for l in range(64)])
    err = np.abs(np.abs(out) - np.abs(ref)).max() / np.abs(ref).max()
    if verbose:
        print(f"S={S} R1={R1} M={M} slots={SL} region {RS * 8} B, workgroup {4 * RS * 8} B: max rel err {err:.2e}; "
              f"worst bank conflict per access class: {conflicts}")
    assert err < 1e-12
    assert all(v == 1 for v in conflicts.values()), conflicts  # every LDS access class conflict-free
    return err


if __name__ == "__main__":
    for S in ([int(a) for a in sys.argv[1:]] or [768, 256, 512, 1024]):
        run(S)
