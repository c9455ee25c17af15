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
