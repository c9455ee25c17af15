"""``acqResults = acquisition(longSignal, settings)`` -- host mirror of
BDS-3_B2a/acquisition.m:1, BDS-3_B1C/acquisition.m:1 and BDS-3_B1C/GPU_acquisition.m:1.

Same arguments, same result fields (``carrFreq``, ``codePhase``, ``peakMetric``: 1 x
max(acqSatelliteList), zeros where not searched / not detected), same console line
``(19 20 . )``.  The work is done by libbds_mi355x.so; this file only converts types.
"""
from __future__ import annotations

import sys
from types import SimpleNamespace

import numpy as np

from . import native

_ctx = {}


def get_context(device: int = 0) -> native.Context:
    """Process-wide bds_ctx per device (keeps the code-spectrum cache warm across calls)."""
    if device not in _ctx:
        _ctx[device] = native.Context(device)
    return _ctx[device]


def release_context(device: int = 0) -> None:
    """Destroy the process-wide context of a device (its HBM -- IF block, code spectra, the inter-pass buffer of the search --
    goes back to the device); the next call builds a new one."""
    c = _ctx.pop(device, None)
    if c is not None:
        c.close()


class AcqResults(SimpleNamespace):
    """acqResults struct (B2a/acquisition.m:161-165)."""


def _as_int8(long_signal, settings):
    a = np.asarray(long_signal)
    if np.iscomplexobj(a):
        # fileType 2: data = I + 1i*Q (B2a/postProcessing.m:92-96)
        inter = np.empty(a.size * 2, dtype=np.float64)
        inter[0::2] = a.real
        inter[1::2] = a.imag
        a, is_complex = inter, True
    else:
        is_complex = False
    if a.dtype != np.int8:
        r = np.rint(a)
        if not np.array_equal(r, a) or r.min() < -128 or r.max() > 127:
            raise ValueError("longSignal must hold int8 values (fread(...,'schar'), postProcessing.m:89-90)")
        a = r.astype(np.int8)
    return np.ascontiguousarray(a).reshape(-1), is_complex


def acquisition(long_signal, settings, device: int = 0, prn_list=None, verbose: bool = True) -> AcqResults:
    """Parallel code-phase search acquisition on the GPU.

    prn_list (extension): the PRN shard this rank searches; results are zero outside
    the shard so an all-reduce(SUM) over ranks reassembles acqResults.
    """
    ctx = get_context(device)
    samples, is_complex = _as_int8(long_signal, settings)
    ctx.acq_load(settings, samples, is_complex)
    ctx.acq_prepare(settings)
    carr, cph, pm, det = ctx.acq_run(settings, prn_list)
    if verbose:  # B2a/acquisition.m:167,259,360,366
        sats = [int(p) for p in (np.atleast_1d(settings.acqSatelliteList) if prn_list is None else prn_list)]
        sys.stdout.write("(" + "".join(f"{p:02d} " if det[p - 1] else ". " for p in sats) + ")\n")
    return AcqResults(carrFreq=carr, codePhase=cph, peakMetric=pm)


def GPU_acquisition(long_signal, settings, **kw) -> AcqResults:
    """B1C/GPU_acquisition.m:1 -- same native entry as acquisition()."""
    return acquisition(long_signal, settings, **kw)
