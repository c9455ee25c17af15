"""C ABI surface and host-side logic (no GPU): the shared library loads, exports every symbol the
header declares, its structs match the ctypes mirror, and the host helpers agree with the oracle."""
import ctypes
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest

import bds_amd
from bds_amd import native
from bds_amd.tracking import field_set
from oracle import tracking as otrk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "bds_mi355x.h")).read()
    return sorted(set(re.findall(r"BDS_API\s+[\w\s\*]+?\b(bds_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = native.lib()
    names = header_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(native.EXPORTS) == names


def test_struct_layout_matches_library():
    lib = native.lib()
    assert lib.bds_abi_check(ctypes.sizeof(native.Settings), ctypes.sizeof(native.Channel),
                             ctypes.sizeof(native.TrackOut), ctypes.sizeof(native.Timing)) == 0
    assert lib.bds_abi_check(1, 2, 3, 4) != 0


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.BdsError, match="no CPU fallback"):
        native.Context(0)


def test_settings_mirror_reference_defaults():
    b1c, b2a = bds_amd.init_settings_b1c(), bds_amd.init_settings_b2a()
    # BDS-3_B1C/initSettings.m:56-57,65,68,80,102-106,116-125 ; BDS-3_B2a/initSettings.m:44-130
    assert (b1c.samplingFreq, b1c.IF, b1c.acqSatelliteList, b1c.numberOfChannels, b1c.msToProcess) == (53e6, 14.58e6, [19, 20], 10, 37000)
    assert (b1c.acqStep, b1c.acqThreshold, b1c.acqCohT, b1c.dllCorrelatorSpacing, b1c.intTime, b1c.CNoInterval) == (50, 7.5, 10, 0.06, 0.01, 50)
    assert (b2a.samplingFreq, b2a.IF, b2a.numberOfChannels, b2a.msToProcess, b2a.fineNoncoh) == (99.375e6, 13.55e6, 12, 49000, 15)
    assert (b2a.acqStep, b2a.acqThreshold, b2a.dllCorrelatorSpacing, b2a.pllNoiseBandwidth, b2a.CNoInterval) == (400, 1.5, 0.5, 20, 200)
    assert bds_amd.init_settings_b1c(acqCohT=5).acqStep == 100  # 1000/acqCohT/2


def test_pack_settings_errors_name_the_field():
    s = bds_amd.init_settings_b2a()
    del s.samplingFreq
    with pytest.raises(AttributeError, match="samplingFreq"):
        native.pack_settings(s)
    with pytest.raises(ValueError, match="signal"):
        native.pack_settings(SimpleNamespace(signal="GPS"))
    cs = native.pack_settings(bds_amd.init_settings_b1c(acqSatelliteList=[3, 7, 63]))
    assert cs.n_acq == 3 and list(cs.acqSatelliteList)[:3] == [3, 7, 63] and cs.signal == 1


def test_long_signal_must_be_int8_valued():
    from bds_amd.acquisition import _as_int8

    s = bds_amd.init_settings_b2a()
    a, cplx = _as_int8(np.array([1.0, -128.0, 127.0]), s)
    assert a.dtype == np.int8 and not cplx
    with pytest.raises(ValueError):
        _as_int8(np.array([0.5, 1.0]), s)
    with pytest.raises(ValueError):
        _as_int8(np.array([300.0]), s)
    a, cplx = _as_int8(np.array([1 + 2j, 3 - 4j]), s)
    assert cplx and list(a) == [1, 2, 3, -4]  # fileType 2 interleave (postProcessing.m:92-96)


def test_field_sets_follow_the_reference_templates():
    b2a = bds_amd.init_settings_b2a(msToProcess=1000)
    n, m, ep, cn, pilot = field_set(b2a, "B2A")
    assert (n, m, pilot) == (1000, 5, True) and "Pilot_I_P" in ep and "Pilot_I_E" not in ep and "SigCNo" in cn
    n, m, ep, cn, pilot = field_set(bds_amd.init_settings_b2a(msToProcess=1000, pilotTRKflag=0), "B2A")
    assert not pilot and "Pilot_I_P" not in ep and cn == ["DataCNo", "DataPLD"]
    b1c = bds_amd.init_settings_b1c(msToProcess=36000)
    n, m, ep, cn, pilot = field_set(b1c, "WB")
    assert (n, m) == (3600, 72) and "Pilot_Q_L" in ep
    n, m, ep, cn, pilot = field_set(bds_amd.init_settings_b1c(msToProcess=36000, pilotTRKflag=1), "NB")
    assert pilot and "Pilot_I_P" in ep and "Pilot_I_E" not in ep


def test_host_helpers_match_oracle():
    s1 = bds_amd.init_settings_b1c()
    np.testing.assert_allclose(native.calc_loop_coef(1, 0.7, 1.0), otrk.calc_loop_coef(1, 0.7, 1.0), rtol=1e-15)
    np.testing.assert_allclose(native.calc_loop_coef_carr(s1), otrk.calc_loop_coef_carr(s1), rtol=1e-15)
    for febw in (27e6, 10e6, 4e6):
        s = s1.copy(FEBW=febw)
        assert abs(native.calc_weighing_factor(s) - otrk.calc_weighing_factor(s)) < 1e-9


def test_resampling_plan_and_fir1_match_oracle():
    """Host side of the resampling branch (acquisition.m:54-124): rate / IF choice and the fir1 taps
    against the oracle's scipy restatement; the filter itself is checked on the GPU."""
    import scipy.signal as ssig

    from oracle import acquisition as oacq

    assert native.resample_plan(bds_amd.init_settings_b2a()) is None  # resamplingflag = 0 (initSettings.m:89)
    for s in (bds_amd.init_settings_b2a(resamplingflag=1),
              bds_amd.init_settings_b1c(resamplingflag=1),
              bds_amd.init_settings_b1c(samplingFreq=40e6, IF=10e6, resamplingflag=1, resamplingThreshold=15e6)):
        fs, fi, wp = native.resample_plan(s)
        x = np.zeros(6000)
        _, s2, old = oacq.resample_condition(x, s)
        assert (fs, fi) == (s2.samplingFreq, s2.IF) and old == (s.samplingFreq, s.IF)
        b = native.fir1_bandpass(701, *wp)
        ref = ssig.firwin(701, list(wp), window="hamming", pass_zero=False, scale=True)
        np.testing.assert_allclose(b, ref, rtol=0, atol=2e-16 * np.abs(ref).max() * 701)
        np.testing.assert_allclose(b, b[::-1], rtol=0, atol=1e-16)  # linear phase
    assert native.resample_plan(bds_amd.init_settings_b1c(samplingFreq=12.5e6, resamplingflag=1)) is None  # fs below threshold


def test_pre_run_matches_oracle():
    rng = np.random.default_rng(0)
    for sig, mk in (("B1C", bds_amd.init_settings_b1c), ("B2A", bds_amd.init_settings_b2a)):
        s = mk(numberOfChannels=6, acqSatelliteList=list(range(1, 21)))
        pm = rng.uniform(1, 30, 20)
        carr = np.where(pm > 12, s.IF + rng.integers(-200, 200, 20) * 25.0, 0.0)
        acq = SimpleNamespace(carrFreq=carr, codePhase=np.where(carr != 0, rng.integers(1, 90000, 20), 0).astype(float), peakMetric=pm)
        ref = otrk.pre_run(acq, s)
        got = bds_amd.pre_run(acq, s)
        for a, b in zip(ref, got):
            assert (a.PRN, a.status, a.acquiredFreq, a.codePhase, a.codeFreq) == (b.PRN, b.status, b.acquiredFreq, b.codePhase, b.codeFreq)


def test_unsupported_branches_are_reported():
    lib = native.lib()
    assert lib.bds_gen_code(1, 2, 64, None, 0) < 0
    out = np.zeros(10, dtype=np.int8)
    assert lib.bds_gen_code(2, 4, 1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), 10) < 0  # BOC61 is B1C only


def test_settings_fields_are_required_not_defaulted():
    """a settings struct with a missing (or misspelt) field is an error naming the
    field -- for the fields of the selected receiver only (B2a has no acqCohT, B1C no fineNoncoh)."""
    import pytest

    s2, s1 = bds_amd.init_settings_b2a(), bds_amd.init_settings_b1c()
    native.pack_settings(s2), native.pack_settings(s1)
    for s, fields in ((s2, ("dllNoiseBandwidth", "acqSearchBand", "fineNoncoh", "dataType", "carrFreqBasis", "CNoInterval")),
                      (s1, ("acqCohT", "pilotACQflag", "FEBW", "pllNoiseBandwidth", "resamplingflag", "skipNumberOfBytes"))):
        for f in fields:
            d = dict(s.__dict__)
            d.pop(f)
            d[f + "x"] = 1  # the misspelt twin
            with pytest.raises(AttributeError, match=f"settings.{f} is missing"):
                native.pack_settings(bds_amd.Settings(**d))
    # fields of the OTHER receiver are not required
    d = dict(s2.__dict__)
    assert "acqCohT" not in d and "FEBW" not in d
    assert native.pack_settings(s2.copy(dataType="int16")).dataType == 1  # rejected by the library (BDS_ERR_UNSUPPORTED)


def test_context_mirror_has_a_method_per_entry_it_wraps():
    """the ctypes mirror keeps one method per native entry the tests and tools call (an edit that drops one -- timing() once --
    must fail here, on CPU, not on the GPU box)"""
    for m in ("acq_load", "acq_prepare", "acq_run", "acq_grid", "acq_peaks", "acq_candidates", "acq_coherent_sums", "timing",
              "track", "reload_tuning", "device_name"):
        assert callable(getattr(native.Context, m, None)), m
