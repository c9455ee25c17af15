"""Parallel code-phase search acquisition (oracle; test infrastructure).

Restates, in float64 / complex128,
  B2a/acquisition.m:126-336   plus the optional resampling pre-conditioner :56-124 and
                              the result recovery :339-356 (off at the shipped
                              configs, B2a/initSettings.m:89)
  B1C/acquisition.m:125-307   (likewise :56-123, :311-328)
The pre-conditioner's fir1 / filtfilt are MATLAB Signal Processing Toolbox calls; they are
restated with scipy.signal.firwin (Hamming window, unit gain at the centre of the pass
band -- fir1's default scaling) and scipy.signal.filtfilt with MATLAB's edge length
3*(numel(b)-1) (odd reflection, steady-state initial conditions).

The D x N ``results`` matrix of the reference (B2a/acquisition.m:154,
B1C/acquisition.m:154; 3.2 GB at the B1C config) is streamed one Doppler row at
a time: the running column maximum, the per-row maximum and the best row are
what lines :218-249 (B2a) / :229-232 (B1C) consume, so the outputs are the same
numbers the materialised matrix would give.

All indices returned are MATLAB 1-based, as in the reference's acqResults.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np
import scipy.fft as sfft

from . import codes
from .matlab import m_round, m_var

_WORKERS = int(os.environ.get("BDS_ORACLE_FFT_WORKERS", "1"))


def _fft(x):
    return sfft.fft(x, workers=_WORKERS)


def _ifft(x):
    return sfft.ifft(x, workers=_WORKERS)


def _new_results(settings):
    n = int(max(settings.acqSatelliteList))
    return SimpleNamespace(
        carrFreq=np.zeros(n), codePhase=np.zeros(n), peakMetric=np.zeros(n)
    )


def resample_condition(long_signal, settings):
    """Input conditioning of B2a/acquisition.m:56-124 / B1C/acquisition.m:56-123.

    Returns (longSignal', settings', old) -- ``old`` is None when the branch is not taken,
    else (oldFreq, oldIF) for :func:`resample_recover`."""
    if not (settings.samplingFreq > settings.resamplingThreshold and int(settings.resamplingflag) == 1):
        return np.asarray(long_signal), settings, None
    import scipy.signal as ssig

    fs = float(settings.samplingFreq)
    IF = float(settings.IF)
    if str(settings.signal).upper() == "B1C":
        bw = 9e6  # B1C :62
    else:
        bw = settings.codeFreqBasis * 2 + 0.5e6  # B2a :62
    w1 = IF - bw / 2
    w2 = IF + bw / 2
    wp = [w1 * 2 / fs - 0.002, w2 * 2 / fs + 0.002]  # :66
    b = ssig.firwin(701, wp, window="hamming", pass_zero=False, scale=True)  # fir1(700, wp) :68
    x = np.asarray(long_signal)
    x = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
    x = ssig.filtfilt(b, [1.0], x, padtype="odd", padlen=3 * 700)  # :70
    fu = IF + bw / 2  # :77
    n = int(np.floor(fu / bw))
    if n < 1:
        n = 1
    lower = 2 * fu / n
    fl = IF - bw / 2
    upper = 2 * fl / (n - 1) if n > 1 else lower
    old_freq = fs
    new_fs = float(np.ceil((lower + upper) / 2))  # :103
    sig_len = int(np.floor((x.size - 1) / old_freq * new_fs))  # :107
    idx = np.ceil(np.arange(sig_len, dtype=np.float64) / new_fs * old_freq).astype(np.int64)  # :109
    idx[0] = 1
    x = x[idx - 1]
    new_if = float(np.fmod(IF, new_fs))  # :119 rem()
    return x, settings.copy(samplingFreq=new_fs, IF=new_if), (old_freq, IF)


def resample_recover(acq, prn, code_phase, settings, old):
    """B2a/acquisition.m:339-356 / B1C/acquisition.m:311-328 (settings = the resampled ones)."""
    old_freq, old_if = old
    acq.codePhase[prn - 1] = np.floor((code_phase - 1) / settings.samplingFreq * old_freq) + 1
    if settings.IF >= settings.samplingFreq / 2:
        doppler = (settings.samplingFreq - settings.IF) - acq.carrFreq[prn - 1]
    else:
        doppler = acq.carrFreq[prn - 1] - settings.IF
    acq.carrFreq[prn - 1] = doppler + old_if


def freq_bins(settings) -> np.ndarray:
    """frqBins(b) = IF - acqSearchBand + acqStep*(b-1), b = 1..D
    (B2a/acquisition.m:150,190-191; B1C/acquisition.m:147,194-195)."""
    d = int(m_round(settings.acqSearchBand * 2 / settings.acqStep)) + 1
    return settings.IF - settings.acqSearchBand + settings.acqStep * np.arange(d, dtype=np.float64)


# ------------------------------------------------------------------------------
# B2a
# ------------------------------------------------------------------------------
def b2a_coarse_rows(long_signal, settings, prn, bins=None):
    """Generator over (bin_index0, row) for one PRN: row = results(bin,:) of
    B2a/acquisition.m:187-211."""
    spc = codes.samples_per_code(settings)
    n = spc * 2
    sig = np.asarray(long_signal[:n])
    ts = 1.0 / settings.samplingFreq
    phase_points = np.arange(n, dtype=np.float64) * 2 * np.pi * ts  # :146
    data_tab = codes.make_b2a_data_table(prn, settings)
    pilot_tab = codes.make_b2a_pilot_table(prn, settings)
    cd = np.conj(_fft(np.concatenate([data_tab, np.zeros(spc)])))  # :179-183
    cp = np.conj(_fft(np.concatenate([pilot_tab, np.zeros(spc)])))
    frq = freq_bins(settings)
    for b in (range(len(frq)) if bins is None else bins):
        carr = np.exp(1j * frq[b] * phase_points)  # :194
        x = _fft(carr * sig)  # :197-201 (real()+1i*imag() of the same product)
        yield b, np.abs(_ifft(x * cd)) + np.abs(_ifft(x * cp))  # :204-209


def acquisition_b2a(long_signal, settings, diag=None, coarse=None):
    """acqResults = acquisition(longSignal, settings)   (B2a/acquisition.m:1).

    ``diag`` (optional dict) receives per-PRN intermediate values for tests.
    ``coarse`` (optional): another evaluator of the Doppler rows -- ``coarse(long_signal, settings, prn)`` returning
    ``(row_max, row_arg0, col_max, row_of)`` with ``row_of(b)`` = results(b+1, :) -- e.g. the C restatement
    (``oracle.cfast.backend()``); everything after the rows stays this function's.
    """
    long_signal, settings, old = resample_condition(long_signal, settings)
    spc = codes.samples_per_code(settings)
    n = spc * 2  # len2ms :134
    samples2chip = int(np.ceil(settings.samplingFreq / settings.codeFreqBasis)) * 2  # :137
    ts = 1.0 / settings.samplingFreq
    frq = freq_bins(settings)
    acq = _new_results(settings)

    for prn in settings.acqSatelliteList:
        prn = int(prn)
        row_max = np.full(len(frq), -np.inf)
        row_arg = np.zeros(len(frq), dtype=np.int64)
        col_max = np.full(n, -np.inf)
        best_row = None
        if coarse is not None:
            row_max, row_arg, col_max, row_of = coarse(long_signal, settings, prn)
        else:
            for b, row in b2a_coarse_rows(long_signal, settings, prn):
                row_arg[b] = int(np.argmax(row))
                row_max[b] = row[row_arg[b]]
                np.maximum(col_max, row, out=col_max)
                if best_row is None or row_max[b] > best_row[0]:
                    best_row = (row_max[b], b, row)
        # :218-221   max(max(results,[],2)) -> first row; max(max(results)) -> first column
        fbin = int(np.argmax(row_max))  # 0-based
        code_phase = int(np.argmax(col_max)) + 1  # 1-based
        peak = float(col_max[code_phase - 1])
        if coarse is not None:
            row = row_of(fbin)
        else:
            row = best_row[2]
            assert best_row[1] == fbin
        # :224-249  second peak in the same bin
        e1 = code_phase - samples2chip
        e2 = code_phase + samples2chip
        e3 = code_phase - spc + samples2chip
        e4 = code_phase + spc - samples2chip
        left = np.arange(max(1, e3), e1 + 1) if e1 >= 1 else np.arange(0)
        right = np.arange(e2, min(e4, n) + 1) if e2 < n else np.arange(0)
        rng = np.concatenate([left, right]).astype(np.int64)
        second = float(np.max(row[rng - 1]))
        acq.peakMetric[prn - 1] = peak / second  # :252
        if diag is not None:
            diag[prn] = dict(peak=peak, second=second, fbin=fbin + 1, codePhase=code_phase,
                             row_max=row_max.copy(), row_arg=row_arg + 1)
        if peak / second > settings.acqThreshold:  # :255
            nfine = int(m_round(settings.acqStep / 25)) + 1  # :265
            dcode = codes.generate_b2a_data_code(prn, settings)
            pcode = codes.generate_b2a_pilot_code(prn, settings)
            nn = int(settings.fineNoncoh) * spc
            # :279-284  floor(ts*k / (1/codeFreqBasis)), k = 1..fineNoncoh*spc
            cvi = np.floor((ts * np.arange(1, nn + 1, dtype=np.float64)) /
                           (1.0 / settings.codeFreqBasis)).astype(np.int64)
            cidx = np.fmod(cvi, int(settings.codeLength))
            long_d = dcode[cidx]
            long_p = pcode[cidx]
            fine_phase = np.arange(nn, dtype=np.float64) * 2 * np.pi * ts  # :287
            sig_fine = long_signal[code_phase - 1: code_phase - 1 + nn]  # :290
            if sig_fine.size != nn:
                raise IndexError("longSignal too short for the B2a fine search "
                                 "(B2a/acquisition.m:290 would throw)")
            fine_frq = np.zeros(nfine)
            fine_res = np.zeros(nfine)
            for k in range(nfine):
                fine_frq[k] = frq[fbin] - settings.acqStep / 2 + 25 * k  # :300-301
                carr = np.exp(1j * fine_frq[k] * fine_phase)  # :303
                b1 = (long_d * carr * sig_fine).reshape(int(settings.fineNoncoh), spc).sum(axis=1)
                b2 = (long_p * carr * sig_fine).reshape(int(settings.fineNoncoh), spc).sum(axis=1)
                fine_res[k] = np.sum(np.abs(b1)) + np.sum(np.abs(b2))  # :321
            kmax = int(np.argmax(fine_res))
            acq.carrFreq[prn - 1] = fine_frq[kmax]  # :329
            acq.codePhase[prn - 1] = code_phase  # :330
            if acq.carrFreq[prn - 1] == 0:
                acq.carrFreq[prn - 1] = 1  # :333-335
            if old is not None:
                resample_recover(acq, prn, code_phase, settings, old)  # :339-356
            if diag is not None:
                diag[prn]["fine"] = fine_res
    return acq


# ------------------------------------------------------------------------------
# B1C
# ------------------------------------------------------------------------------
def _b1c_sizes(settings):
    spc = codes.samples_per_code(settings)
    x_len = int(m_round(spc / 10 * settings.acqCohT))  # samplesXmsLen :132
    n = int(m_round(spc / 10 * (10 + settings.acqCohT)))  # len10PlusXms :135
    return spc, x_len, n


def b1c_coarse_rows(long_signal, settings, prn, bins=None):
    """Generator over (bin_index0, row): row = results(bin,:) of B1C/acquisition.m:191-222."""
    spc, x_len, n = _b1c_sizes(settings)
    sig = np.asarray(long_signal[:n])
    ts = 1.0 / settings.samplingFreq
    phase_points = np.arange(n, dtype=np.float64) * 2 * np.pi * ts  # :144
    data_tab = codes.make_data_table(settings, prn)
    cd = np.conj(_fft(np.concatenate([data_tab[:x_len], np.zeros(n - x_len)])))  # :176-180
    if settings.pilotACQflag == 1:
        pilot_tab = codes.make_pilot_table(settings, prn)
        cp = np.conj(_fft(np.concatenate([pilot_tab[:x_len], np.zeros(n - x_len)])))  # :184-187
    frq = freq_bins(settings)
    for b in (range(len(frq)) if bins is None else bins):
        carr = np.exp(1j * frq[b] * phase_points)  # :198
        x = _fft(carr * sig)  # :201-205
        row = np.abs(_ifft(x * cd))  # :209-212
        if settings.pilotACQflag == 1:
            row = (row * np.sqrt(11) + np.abs(_ifft(x * cp)) * np.sqrt(29)) / np.sqrt(40)  # :216-219
        yield b, row


def acquisition_b1c(long_signal, settings, diag=None, coarse=None):
    """acqResults = acquisition(longSignal, settings)   (B1C/acquisition.m:1).  ``coarse``: see acquisition_b2a."""
    long_signal, settings, old = resample_condition(long_signal, settings)
    spc, x_len, n = _b1c_sizes(settings)
    ts = 1.0 / settings.samplingFreq
    frq = freq_bins(settings)
    sig_power = np.sqrt(m_var(long_signal[:x_len]) * x_len)  # :150
    acq = _new_results(settings)

    for prn in settings.acqSatelliteList:
        prn = int(prn)
        row_max = np.full(len(frq), -np.inf)
        row_arg = np.zeros(len(frq), dtype=np.int64)
        col_max = np.full(n, -np.inf)
        if coarse is not None:
            row_max, row_arg, col_max, _ = coarse(long_signal, settings, prn)
        else:
            for b, row in b1c_coarse_rows(long_signal, settings, prn):
                row_arg[b] = int(np.argmax(row))
                row_max[b] = row[row_arg[b]]
                np.maximum(col_max, row, out=col_max)
        fbin = int(np.argmax(row_max))  # :229
        code_phase = int(np.argmax(col_max)) + 1  # :232
        peak = float(col_max[code_phase - 1])
        acq.peakMetric[prn - 1] = peak / sig_power  # :235
        if code_phase + spc - 1 > long_signal.size:  # :239-241
            code_phase -= spc
        if diag is not None:
            diag[prn] = dict(peak=peak, sigPower=sig_power, fbin=fbin + 1, codePhase=code_phase,
                             row_max=row_max.copy(), row_arg=row_arg + 1)
        if peak / sig_power > settings.acqThreshold:  # :244
            data_tab = codes.make_data_table(settings, prn)
            sig0 = np.asarray(long_signal[code_phase - 1: code_phase - 1 + spc])
            sig0 = sig0.astype(np.complex128 if np.iscomplexobj(sig0) else np.float64)
            sig0 = sig0 - np.mean(sig0)  # :253-254
            xc = sig0 * data_tab  # :257
            if settings.pilotACQflag == 1:
                xcp = sig0 * codes.make_pilot_table(settings, prn)  # :261
            nfine = int(m_round(settings.acqStep / 25)) * 2 + 1  # :267
            fine_phase = np.arange(spc, dtype=np.float64) * 2 * np.pi * ts  # :276
            fine_frq = np.zeros(nfine)
            fine_res = np.zeros(nfine)
            for k in range(nfine):
                fine_frq[k] = frq[fbin] - settings.acqStep + 25 * k  # :282-283
                carr = np.exp(1j * fine_frq[k] * fine_phase)  # :285
                fine_res[k] = np.abs(np.sum(xc * carr))  # :287
                if settings.pilotACQflag == 1:
                    fine_res[k] = (fine_res[k] * 11 + np.abs(np.sum(xcp * carr)) * 29) / 40  # :291-292
            kmax = int(np.argmax(fine_res))
            acq.carrFreq[prn - 1] = fine_frq[kmax]  # :300
            if acq.carrFreq[prn - 1] == 0:
                acq.carrFreq[prn - 1] = 1  # :303-305
            acq.codePhase[prn - 1] = code_phase  # :307
            if old is not None:
                resample_recover(acq, prn, code_phase, settings, old)  # :311-328
            if diag is not None:
                diag[prn]["fine"] = fine_res
    return acq


def acquisition(long_signal, settings, diag=None, coarse=None):
    """Dispatch on settings.signal ('B1C' | 'B2A') -- the reference keeps one
    acquisition.m per receiver directory."""
    if str(settings.signal).upper() == "B1C":
        return acquisition_b1c(long_signal, settings, diag, coarse)
    return acquisition_b2a(long_signal, settings, diag, coarse)
