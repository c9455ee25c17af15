"""Frame-sync correlators: oracle properties and host-side patterns on the CPU, GPU parity
(bit-exact, integer correlation) through bds_frame_sync."""
from types import SimpleNamespace

import numpy as np
import pytest

import bds_amd
from bds_amd import native
from oracle import codes, framesync as ofs


def test_secondary_code_properties():
    """generate2ndCode.m:59-84: 1800 chips of a Weil code over N = 3607; distinct PRNs give distinct,
    nearly balanced, weakly cross-correlated sequences; the library generator matches the oracle."""
    seqs = [codes.generate_2nd_code(p) for p in range(1, 64)]
    for p, s in enumerate(seqs, 1):
        assert s.shape == (1800,) and set(np.unique(s)) == {-1.0, 1.0}
        assert abs(s.sum()) < 120
        np.testing.assert_array_equal(native.gen_code("B1C", "pilot_secondary", p), s.astype(np.int8))
        np.testing.assert_array_equal(native.sync_pattern("B1C", p), s.astype(np.int8))
    g = np.stack(seqs) @ np.stack(seqs).T
    assert np.all(np.diag(g) == 1800) and np.abs(g - np.diag(np.diag(g))).max() < 250
    # Legendre sequence over 3607 against the Jacobi symbol the reference evaluates (generate2ndCode.m:63-67)
    leg = codes.legendre_sequence(3607)
    for i in (1, 2, 3, 5, 1000, 1803, 3606):
        assert leg[i] == (1 if codes.jacobi_symbol(i, 3607) == 1 else 0)


def test_b2a_pattern_and_xcorr_second_half():
    pat = ofs.b2a_pattern()
    assert pat.size == 120 and pat[:5].tolist() == [-1, -1, -1, 1, -1]
    np.testing.assert_array_equal(native.sync_pattern("B2A"), pat.astype(np.int8))
    a = np.array([1.0, 2.0, 3.0, 4.0])
    b = np.array([1.0, 1.0])
    # xcorr([1 2 3 4],[1 1]) second half: lag 0: 1+2, lag 1: 2+3, lag 2: 3+4, lag 3: 4
    np.testing.assert_array_equal(ofs.xcorr_second_half(a, b), [3, 5, 7, 4])
    np.testing.assert_array_equal(ofs.xcorr_second_half(b, a), [3, 1, 0, 0])  # shorter first input is padded


def _b1c_prompt(rng, prn, n, start, flips, noise):
    sec = codes.generate_2nd_code(prn)
    x = rng.normal(0, 1, n) * 50
    pos = start
    for f in flips:
        if pos + 1800 <= n:
            x[pos:pos + 1800] = f * 1000.0 * sec
        pos += 1800
    return x + rng.normal(0, noise, n)


@pytest.mark.gpu
def test_b1c_secondary_code_sync_gpu(ctx):
    rng = np.random.default_rng(4)
    prns = [3, 27, 58]
    n = 5000
    prompt = np.stack([_b1c_prompt(rng, p, n, 137 + 11 * i, [1, -1], 40.0) for i, p in enumerate(prns)])
    xc, idx = ctx.frame_sync("B1C", prns, prompt)
    for i, p in enumerate(prns):
        r, ind = ofs.frame_sync_b1c(prompt[i], p)
        np.testing.assert_array_equal(xc[i], r.astype(np.int64))
        np.testing.assert_array_equal(idx[i], ind)
        assert ind.tolist() == [137 + 11 * i + 1, 137 + 11 * i + 1801]
    # fewer epochs than the code length: xcorr pads the bits, nothing can reach 1800
    short = prompt[:, :700]
    xc, idx = ctx.frame_sync("B1C", prns, short)
    assert xc.shape == (3, 1800) and all(len(v) == 0 for v in idx)
    np.testing.assert_array_equal(xc[1], ofs.frame_sync_b1c(short[1], prns[1])[0].astype(np.int64))


@pytest.mark.gpu
def test_b2a_preamble_sync_gpu(ctx):
    rng = np.random.default_rng(5)
    n = 9000
    pat = ofs.b2a_pattern()
    rows = []
    for c in range(4):
        bits = rng.choice([-1.0, 1.0], n // 5 + 1)
        x = np.repeat(bits, 5)[:n] * np.tile(ofs.B2A_SECOND_CODE, n // 5 + 1)[:n]
        for k, start in enumerate((40 + 5 * c, 3040 + 5 * c, 6040 + 5 * c)):
            x[start:start + 120] = pat * (1 if (k + c) % 2 == 0 else -1)
        rows.append(x * 800 + rng.normal(0, 150, n))
    prompt = np.stack(rows)
    xc, idx = ctx.frame_sync("B2A", [5, 9, 19, 33], prompt)
    for c in range(4):
        r, ind = ofs.frame_sync_b2a(prompt[c])
        np.testing.assert_array_equal(xc[c], r.astype(np.int64))
        np.testing.assert_array_equal(idx[c], ind)
        assert {40 + 5 * c + 1, 3040 + 5 * c + 1, 6040 + 5 * c + 1} <= set(ind.tolist())
    # host mirror over trackResults-like objects, PRN 0 channels skipped
    tr = [SimpleNamespace(PRN=5, I_P=prompt[0]), SimpleNamespace(PRN=0, I_P=np.zeros(n)), SimpleNamespace(PRN=9, I_P=prompt[1])]
    out = bds_amd.frame_sync(tr, bds_amd.init_settings_b2a())
    assert len(out) == 2
    np.testing.assert_array_equal(out[1][1], ofs.frame_sync_b2a(prompt[1])[1])
