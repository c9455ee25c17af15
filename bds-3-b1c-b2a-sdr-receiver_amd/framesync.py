"""Frame-synchronisation correlators over trackResults -- the first step of the reference's
navigation decoding that consumes this package's outputs:

    B1C  B1C/include/BCNAV1decoding.m:66-91   secondary-code correlation over Pilot_I_P (wide-band
         tracking, pilotTRKflag == 2) or Pilot_Q_P (narrow-band)
    B2a  B2a/include/BCNAV2decoding.m:69-97   preamble (x) NH correlation over I_P

Returns, per channel, the second half of ``xcorr(bits, pattern)`` and ``index`` (1-based, as
``find`` gives it).  The correlation runs on the GPU (``bds_frame_sync``); no CPU fallback.
"""
from __future__ import annotations

import numpy as np

from .acquisition import get_context


def unpack_cplx(filename_in, filename_out, device: int = 0):
    """unpack_cplx(filename_in, filename_out) -- B2a/include/unpack_cplx.m: packed 2+2-bit I/Q bytes
    to the int8 pairs of a fileType-2 record (device kernel, one 32-bit store per input byte)."""
    get_context(device).unpack_cplx_file(filename_in, filename_out)


def frame_sync(track_results, settings, device: int = 0):
    """[(XcorrResult, index), ...] for every tracked channel (PRN != 0) of ``track_results``."""
    b1c = str(settings.signal).upper() == "B1C"
    chans = [r for r in track_results if int(getattr(r, "PRN", 0)) != 0]
    if not chans:
        return []
    if b1c:
        field = "Pilot_I_P" if int(settings.pilotTRKflag) == 2 else "Pilot_Q_P"  # BCNAV1decoding.m:66-73
    else:
        field = "I_P"  # BCNAV2decoding.m:84
    prompt = np.stack([np.asarray(getattr(r, field), dtype=np.float64) for r in chans])
    xc, idx = get_context(device).frame_sync("B1C" if b1c else "B2A", [int(r.PRN) for r in chans], prompt)
    return list(zip(xc, idx))
