"""ctypes front of the C restatement of the coarse search (oracle/c/acq_oracle.c; TEST INFRASTRUCTURE ONLY).

Same role and same rules as the rest of ``oracle/``: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The library is built by ``make -C oracle/c``
(``__graft_entry__.build()`` does it) into ``oracle/_build/liboracle_acq.so`` -- git-ignored, travels to the GPU
box with the snapshot.  PARITY UNPINNED, like the NumPy restatement it is checked against (tests/test_oracle_c.py).

``coarse_rows`` evaluates the rows ``results(b, :)`` of B1C/acquisition.m:191-222 / B2a/acquisition.m:187-211 for one
PRN with its own float64 mixed-radix transform (OpenMP over the Doppler bins) and returns the reductions the reference
takes from the D x N matrix: row maxima with their first index (``max(results, [], 2)``) and the column maximum
(``max(results)``)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from . import acquisition as _acq
from . import codes as _codes

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_acq.so")
_lib = None


def build() -> str:
    """gcc the C restatement (no-op when up to date); returns the path of the shared object."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return _SO


def available() -> bool:
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f"{_SO} not built: run `make -C oracle/c` (or __graft_entry__.build())")
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        L.bds_oracle_fft.argtypes = [dp, dp, ctypes.c_long, ctypes.c_int]
        L.bds_oracle_fft.restype = ctypes.c_int
        L.bds_oracle_coarse_rows.argtypes = [dp, dp, ctypes.c_long, ctypes.c_double, dp, dp, ctypes.c_long, dp, ctypes.c_int,
                                             ctypes.c_int, dp, ctypes.POINTER(ctypes.c_long), dp, dp, ctypes.c_int]
        L.bds_oracle_coarse_rows.restype = ctypes.c_int
        L.bds_oracle_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def fft(x, inverse=False):
    """fft(x) / ifft(x) with MATLAB's (FFTW's) normalisation, by the C oracle's own transform."""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.complex128))
    out = np.empty_like(a)
    rc = lib().bds_oracle_fft(_dp(a.view(np.float64)), _dp(out.view(np.float64)), a.size, 1 if inverse else 0)
    if rc:
        raise RuntimeError(f"bds_oracle_fft: {rc}")
    return out


def coarse_rows(long_signal, settings, prn, bins=None, want_rows=False, col_max=None, threads=0):
    """Rows of one PRN.  Returns (row_max, row_arg0, col_max, rows | None); ``col_max`` (n doubles) is updated in place when
    given -- pass the same array for successive bin subsets -- else a fresh one is returned."""
    b1c = str(settings.signal).upper() == "B1C"
    if b1c:
        spc, x_len, n = _acq._b1c_sizes(settings)
        cd = _codes.make_data_table(settings, prn)[:x_len]
        cp = _codes.make_pilot_table(settings, prn)[:x_len] if settings.pilotACQflag == 1 else None
        kind = 1
    else:
        spc = _codes.samples_per_code(settings)
        n, x_len = 2 * spc, spc
        cd = _codes.make_b2a_data_table(prn, settings)
        cp = _codes.make_b2a_pilot_table(prn, settings)
        kind = 0
    sig = np.asarray(long_signal[:n])
    if sig.size != n:
        raise IndexError("longSignal shorter than the coarse-search block")
    re = np.ascontiguousarray(sig.real, dtype=np.float64)
    im = np.ascontiguousarray(sig.imag, dtype=np.float64) if np.iscomplexobj(sig) else None
    frq = _acq.freq_bins(settings)
    if bins is not None:
        frq = frq[np.asarray(list(bins), dtype=np.int64)]
    frq = np.ascontiguousarray(frq, dtype=np.float64)
    nb = frq.size
    row_max = np.empty(nb)
    row_arg = np.empty(nb, dtype=np.int64)
    if col_max is None:
        col_max = np.full(n, -np.inf)
    rows = np.empty((nb, n)) if want_rows else None
    cd = np.ascontiguousarray(cd, dtype=np.float64)
    cp = None if cp is None else np.ascontiguousarray(cp, dtype=np.float64)
    rc = lib().bds_oracle_coarse_rows(_dp(re), _dp(im), n, float(settings.samplingFreq), _dp(cd), _dp(cp), x_len, _dp(frq), nb, kind,
                                      _dp(row_max), row_arg.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), _dp(col_max), _dp(rows),
                                      int(threads))
    if rc:
        raise RuntimeError(f"bds_oracle_coarse_rows: {rc}")
    return row_max, row_arg, col_max, rows


def threads() -> int:
    return int(lib().bds_oracle_threads())


def backend(threads=0):
    """The ``coarse=`` evaluator of oracle.acquisition.acquisition_b1c / _b2a on the C restatement: the Doppler rows of a PRN by
    oracle/c/acq_oracle.c, everything after them (peak, second peak / GLRT metric, threshold, fine search) by the NumPy oracle."""

    def coarse(long_signal, settings, prn):
        rm, ra, cm, _ = coarse_rows(long_signal, settings, prn, threads=threads)

        def row_of(b):
            return coarse_rows(long_signal, settings, prn, bins=[b], want_rows=True, threads=1)[3][0]

        return rm, ra, cm, row_of

    return coarse


# ---- tracking: one epoch's sample loop by oracle/c/trk_oracle.c --------------------------------------------------------------------
_code_cache = {}


def _code_f64(a):
    if a is None:
        return None
    key = id(a)
    hit = _code_cache.get(key)
    if hit is None or hit[0] is not a:
        if len(_code_cache) > 256:
            _code_cache.clear()
        hit = (a, np.ascontiguousarray(a, dtype=np.float64))
        _code_cache[key] = hit
    return hit[1]


def trk_epoch(raw8, blk, iq, rem_code, step, spc_el, scale, rem_carr, carr_freq, fs, b2a, dcode, pcode, p6code):
    """The ``correlate=`` evaluator of oracle.tracking.tracking on the C restatement: (sums[18], t_p_last, trig_end)."""
    L = lib()
    if not getattr(L, "_trk_ready", False):
        dp = ctypes.POINTER(ctypes.c_double)
        L.bds_oracle_trk_epoch.argtypes = [ctypes.POINTER(ctypes.c_int8), ctypes.c_long, ctypes.c_int] + [ctypes.c_double] * 7 + [ctypes.c_int, dp, dp, dp,
                                                                                                                                 dp, dp, dp]
        L.bds_oracle_trk_epoch.restype = ctypes.c_int
        L._trk_ready = True
    raw8 = np.ascontiguousarray(raw8, dtype=np.int8)
    sums = np.zeros(18)
    t_p = ctypes.c_double()
    t_e = ctypes.c_double()
    d, p, p6 = _code_f64(dcode), _code_f64(pcode), _code_f64(p6code)
    rc = L.bds_oracle_trk_epoch(raw8.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), int(blk), 1 if iq else 0, float(rem_code), float(step),
                                float(spc_el), float(scale), float(rem_carr), float(carr_freq), float(fs), 1 if b2a else 0, _dp(d), _dp(p), _dp(p6),
                                _dp(sums), ctypes.byref(t_p), ctypes.byref(t_e))
    if rc:
        raise RuntimeError(f"bds_oracle_trk_epoch: {rc}")
    return sums, t_p.value, t_e.value


def tracking_parallel(data, channel, settings, mode=None, threads=None):
    """oracle.tracking.tracking with the sample loops in C, one thread per channel (the channels are independent loops over the same
    record; ctypes releases the interpreter lock inside the C call).  ``data``: the record as an int8 array (np.memmap works).
    Returns the list of per-channel results in channel order (a channel that hits the end of the record keeps status '-', like the
    reference's first such channel; later channels are still run -- the callers here use records that are long enough)."""
    from concurrent.futures import ThreadPoolExecutor

    from . import tracking as _trk

    def one(ch):
        res, _ = _trk.tracking(_trk.RawFile(data), [ch], settings.copy(numberOfChannels=1), mode=mode, correlate=trk_epoch)
        return res[0]

    n = len(channel)
    with ThreadPoolExecutor(max_workers=max(1, min(n, threads or n))) as ex:
        return list(ex.map(one, channel))
