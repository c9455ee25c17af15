"""``[trackResults, channel] = tracking(fid, channel, settings)`` -- host mirror of
BDS-3_B2a/tracking.m:1, BDS-3_B1C/NB_tracking.m:1, BDS-3_B1C/WB_tracking.m:1 and of
``channel = preRun(acqResults, settings)`` (include/preRun.m:1).

``fid`` may be a file path, an open binary file object (its ``.name`` is used: the
reference seeks absolutely from 'bof', B2a/tracking.m:151-153, so the handle position
is irrelevant) or an int8 array holding the raw file bytes.  The result is a list of
per-channel structs with exactly the field set the reference variant creates
(SURVEY.md Appendix D).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np

from . import native
from .acquisition import get_context


class TrackResults(SimpleNamespace):
    """One element of the trackResults struct array."""


def pre_run(acq_results, settings):
    """channel = preRun(acqResults, settings)  (B1C/include/preRun.m, B2a/include/preRun.m)."""
    ch = native.pre_run(settings, acq_results.carrFreq, acq_results.codePhase, acq_results.peakMetric)
    return [SimpleNamespace(PRN=int(c.PRN), acquiredFreq=float(c.acquiredFreq), codePhase=float(c.codePhase),
                            codeFreq=float(c.codeFreq), status=chr(c.status)) for c in ch]


def _mode(settings, mode):
    if mode is None:
        if str(settings.signal).upper() == "B2A":
            return "B2A"
        return "WB" if int(settings.pilotTRKflag) == 2 else "NB"  # B1C/postProcessing.m:137-143
    return mode


def _round_half_away(x):
    return int(np.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def field_set(settings, mode):
    """(n_epochs, n_cno, per-epoch fields, C/N0 fields) of the reference template
    (B2a/tracking.m:48-93, NB_tracking.m:53-102, WB_tracking.m:53-109)."""
    if mode == "B2A":
        n = int(settings.msToProcess)
        pilot = int(settings.pilotTRKflag) == 1
    else:
        n = _round_half_away(settings.msToProcess / 1000 / settings.intTime)
        pilot = int(settings.pilotTRKflag) == (2 if mode == "WB" else 1)
    m = n // int(settings.CNoInterval)
    ep = ["absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L"]
    if pilot:
        ep += ["Pilot_I_P", "Pilot_Q_P"]
        if mode == "WB":
            ep += ["Pilot_I_E", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_L"]
    ep += ["dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt", "remCodePhase", "remCarrPhase"]
    cn = ["DataCNo", "DataPLD"] + (["PilotCNo", "PilotPLD", "SigCNo"] if pilot else [])
    return n, m, ep, cn, pilot


def tracking(fid, channel, settings, mode=None, device: int = 0):
    mode = _mode(settings, mode)
    s = settings.copy() if hasattr(settings, "copy") else settings
    if mode in ("NB", "WB") and str(settings.signal).upper() != "B1C":
        raise ValueError("NB/WB tracking are B1C variants")
    n, m, ep, cn, pilot = field_set(settings, mode)
    if isinstance(fid, (str, bytes, os.PathLike)):
        source = fid
    elif hasattr(fid, "name") and not isinstance(fid, np.ndarray):
        source = fid.name
    else:
        source = np.ascontiguousarray(fid, dtype=np.int8)
    ctx = get_context(device)
    # the native side derives the variant from settings.signal / pilotTRKflag exactly as
    # postProcessing.m does; an explicit NB request on a pilotTRKflag==2 struct is honoured
    # by passing the flag the variant tests for
    if mode == "NB" and int(s.pilotTRKflag) == 2:
        s = settings.copy(pilotTRKflag=0)
    if mode == "WB" and int(s.pilotTRKflag) != 2:  # WB_tracking.m:78 only tests == 2
        s = settings.copy(pilotTRKflag=0)
    arr = ctx.track(s, source, channel, n, m, ep + cn)
    sig_name = "B2a_CNo" if mode == "B2A" else "B1C_CNo"
    out = []
    for c in range(len(channel)):
        r = TrackResults()
        r.status = chr(int(arr["status"][c])) if arr["status"][c] else "-"
        for f in ep:
            setattr(r, f, arr[f][c].copy())
        for f in cn:
            setattr(r, sig_name if f == "SigCNo" else f, arr[f][c].copy())
        r.PRN = int(channel[c].PRN) if int(channel[c].PRN) != 0 else None  # lazily added field, tracking.m:144
        r.completed = int(arr["completed"][c])
        out.append(r)
    return out, channel


def _results(arr, channel_prns, n_ch, ep, cn, mode):
    sig_name = "B2a_CNo" if mode == "B2A" else "B1C_CNo"
    out = []
    for c in range(n_ch):
        r = TrackResults()
        r.status = chr(int(arr["status"][c])) if arr["status"][c] else "-"
        for f in ep:
            setattr(r, f, arr[f][c].copy())
        for f in cn:
            setattr(r, sig_name if f == "SigCNo" else f, arr[f][c].copy())
        r.PRN = int(channel_prns[c]) if int(channel_prns[c]) != 0 else None
        r.completed = int(arr["completed"][c])
        out.append(r)
    return out


def acquire_track(long_signal, path, settings, device: int = 0):
    """The acquisition -> preRun -> tracking section of postProcessing.m (B2a/postProcessing.m:100-123,
    B1C/postProcessing.m:105-143) as ONE native call: bds_acquire_track runs the search, allocates the channels with a
    device kernel (bds_pre_run_device) and tracks the record at `path` with the variant the settings select, without
    returning to the host language in between.  Returns (acqResults, channel, trackResults)."""
    mode = _mode(settings, None)
    n, m, ep, cn, pilot = field_set(settings, mode)
    x = np.asarray(long_signal)
    is_complex = np.iscomplexobj(x)
    if is_complex:  # fileType 2: interleaved int8 pairs, as acquisition() hands them over
        pairs = np.empty(2 * x.size, dtype=np.int8)
        pairs[0::2], pairs[1::2] = x.real.astype(np.int8), x.imag.astype(np.int8)
        x = pairs
    ctx = get_context(device)
    (carr, cph, pm, det), ch, arr = ctx.acquire_track(settings, np.ascontiguousarray(x, dtype=np.int8), is_complex, path, n, m, ep + cn)
    acq = SimpleNamespace(carrFreq=carr, codePhase=cph, peakMetric=pm)
    channel = [SimpleNamespace(PRN=int(c.PRN), acquiredFreq=float(c.acquiredFreq), codePhase=float(c.codePhase),
                               codeFreq=float(c.codeFreq), status=chr(c.status)) for c in ch]
    return acq, channel, _results(arr, [c.PRN for c in channel], len(channel), ep, cn, mode)


def NB_tracking(fid, channel, settings, **kw):
    """B1C/NB_tracking.m:1."""
    return tracking(fid, channel, settings, mode="NB", **kw)


def WB_tracking(fid, channel, settings, **kw):
    """B1C/WB_tracking.m:1."""
    return tracking(fid, channel, settings, mode="WB", **kw)
