"""Packed-sample converter (B2a/include/unpack_cplx.m): the oracle against the reference's own literal
look-up tables (tests/golden/unpack_cplx_lut.npz, extracted by tests/golden/make_unpack_lut.py), and the
device kernel against both."""
import os

import numpy as np
import pytest

import bds_amd
from oracle import unpack as oun

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_matches_the_reference_tables():
    lut = np.load(os.path.join(GOLD, "unpack_cplx_lut.npz"))["lut"]
    assert lut.shape == (256, 4) and set(np.unique(lut)) == {-3, -1, 1, 3}
    np.testing.assert_array_equal(oun.unpack_cplx(np.arange(256, dtype=np.uint8)).reshape(256, 4), lut)
    assert oun.unpack_cplx(np.zeros(0, dtype=np.uint8)).size == 0


@pytest.mark.gpu
def test_device_unpack_is_bit_exact(ctx, tmp_path):
    lut = np.load(os.path.join(GOLD, "unpack_cplx_lut.npz"))["lut"]
    np.testing.assert_array_equal(ctx.unpack_cplx(np.arange(256, dtype=np.uint8)).reshape(256, 4), lut)
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, 3_000_001, dtype=np.uint8)  # ragged length, several grid strides
    np.testing.assert_array_equal(ctx.unpack_cplx(data), oun.unpack_cplx(data))
    fin, fout = tmp_path / "packed.bin", tmp_path / "iq.bin"
    data[:100_003].tofile(fin)
    bds_amd.unpack_cplx(str(fin), str(fout))
    np.testing.assert_array_equal(np.fromfile(fout, dtype=np.int8), oun.unpack_cplx(data[:100_003]))
    with pytest.raises(bds_amd.native.BdsError, match="Unable to read"):
        bds_amd.unpack_cplx(str(tmp_path / "missing.bin"), str(fout))
