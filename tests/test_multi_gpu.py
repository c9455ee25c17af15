"""bds_acquire_multi (one host process, N devices, RCCL inside the library) on the devices this box has: with one
device the partition gives it every job and the result must equal bds_acquire bit for bit; the all-reduce is
still taken through RCCL (communicator of every device of the context) via the library's test hook.  Two signals
in one call = the BASELINE.json configs[4] shape (B1C + B2a jointly) at reduced size."""
import numpy as np
import pytest

import bds_amd
from bds_amd import native

from helpers import medium_b2a, small_b1c

pytestmark = pytest.mark.gpu


def test_multi_equals_single_device_and_goes_through_rccl(ctx, monkeypatch):
    s2, x2, _ = medium_b2a()
    s1, x1, _ = small_b1c()
    want2 = bds_amd.acquisition(x2, s2, verbose=False)
    want1 = bds_amd.acquisition(x1, s1, verbose=False)
    monkeypatch.setenv("BDS_MULTI_FORCE_RCCL", "1")  # a single device has nothing to exchange; go through RCCL anyway
    m = native.MultiContext()  # every visible device
    try:
        n = m.size()
        assert n >= 1
        (c1, p1, m1, d1), (c2, p2, m2, d2) = m.acquire([(s1, x1, False), (s2, x2, False)])
        assert m.rccl_ranks() == n
    finally:
        m.close()
    for got, want in (((c1, p1, m1), want1), ((c2, p2, m2), want2)):
        np.testing.assert_array_equal(got[0], want.carrFreq)
        np.testing.assert_array_equal(got[1], want.codePhase)
        np.testing.assert_array_equal(got[2], want.peakMetric)
    assert d1[2] == 1 and d1[6] == 0 and set(np.nonzero(d2)[0] + 1) == {9, 19}


def test_multi_argument_errors():
    with pytest.raises(native.BdsError, match="listed twice"):
        native.MultiContext([0, 0])
    m = native.MultiContext([0])
    try:
        s2, x2, _ = medium_b2a()
        with pytest.raises(native.BdsError, match="acquisition needs at least"):
            m.acquire([(s2, x2[:1000], False)])
    finally:
        m.close()


def _same(got, want):
    np.testing.assert_array_equal(got[0], want.carrFreq)
    np.testing.assert_array_equal(got[1], want.codePhase)
    np.testing.assert_array_equal(got[2], want.peakMetric)


def test_two_contexts_aliased_onto_one_device(ctx, monkeypatch):
    """The N > 1 path of bds_acquire_multi on a one-GPU box (BDS_MULTI_TEST_ALIAS): two contexts on device 0, the LPT
    partition of the (signal, PRN) jobs over them, one host thread per context running load / prepare / run
    concurrently, zero-filled partial results, and their sum -- taken on the host here, by RCCL on a real node --
    must equal the single-device results bit for bit (x + 0)."""
    s2, x2, _ = medium_b2a()
    s1, x1, _ = small_b1c()
    want2 = bds_amd.acquisition(x2, s2, verbose=False)
    want1 = bds_amd.acquisition(x1, s1, verbose=False)
    monkeypatch.setenv("BDS_MULTI_TEST_ALIAS", "1")
    for ids in ([0, 0], [0, 0, 0]):
        m = native.MultiContext(ids)
        try:
            assert m.size() == len(ids)
            (c1, p1, m1, d1), (c2, p2, m2, d2) = m.acquire([(s1, x1, False), (s2, x2, False)])
            # and again: cached code spectra / plans of both contexts
            (e1, q1, n1, _), (e2, q2, n2, _) = m.acquire([(s1, x1, False), (s2, x2, False)])
        finally:
            m.close()
        _same((c1, p1, m1), want1)
        _same((c2, p2, m2), want2)
        _same((e1, q1, n1), want1)
        _same((e2, q2, n2), want2)
        assert d1[2] == 1 and d1[6] == 0 and set(np.nonzero(d2)[0] + 1) == {9, 19}


def test_missing_rccl_is_a_clean_error(ctx, monkeypatch):
    """A node whose RCCL cannot be loaded must get BDS_ERR_UNSUPPORTED with the loader's message, not a crash
    (dlerror() is read once; BDS_RCCL_LIB overrides the library name)."""
    s2, x2, _ = medium_b2a()
    monkeypatch.setenv("BDS_MULTI_FORCE_RCCL", "1")
    monkeypatch.setenv("BDS_RCCL_LIB", "/nonexistent/librccl_not_here.so")
    m = native.MultiContext([0])
    try:
        with pytest.raises(native.BdsError, match="cannot load RCCL: .*librccl_not_here"):
            m.acquire([(s2, x2, False)])
    finally:
        m.close()


def test_two_real_devices(ctx):
    """Two physical devices, RCCL all-reduce between them (skipped on the one-GPU boxes of this build's pool)."""
    import ctypes as C

    n = C.c_int(0)
    hip = C.CDLL("libamdhip64.so")
    hip.hipGetDeviceCount(C.byref(n))
    if n.value < 2:
        pytest.skip("needs two GPUs")
    s2, x2, _ = medium_b2a()
    want2 = bds_amd.acquisition(x2, s2, verbose=False)
    m = native.MultiContext([0, 1])
    try:
        ((c2, p2, m2, d2),) = m.acquire([(s2, x2, False)])
        assert m.rccl_ranks() == 2
    finally:
        m.close()
    _same((c2, p2, m2), want2)
