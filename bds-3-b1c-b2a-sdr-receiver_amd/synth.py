"""Seeded synthetic int8 IF generator (bench / test input).

The reference ships no IF recording (README.md:91-148 lists downloads only), so
every workload here is synthetic.  The signal is written in the reference's own
code arrays and carrier conventions so that its loops lock with the documented
signs:

  B1C:  x = s_I cos(th) - s_Q sin(th),
        s_I = 1/2 D dataBOC11 - sqrt(1/11) pilotBOC61,  s_Q = sqrt(29/44) pilotBOC11 S
        (WB_tracking.m:375-380 then adds the BOC(6,1) and BOC(1,1) pilot parts
        constructively; NB_tracking.m:357 sees the pilot on Q)
  B2a:  x = D c_d sin(th) + S c_p cos(th)
        (tracking.m:309-314: data power lands in I_P = sum c imag(e^{+j th} x))

  th = 2 pi (IF + f_d) t + phi0; code rate scaled by (1 + f_d / carrFreqBasis);
  D / S = random +-1 per primary-code period; noise N(0, sigma); round + clip
  to int8.  Amplitude per satellite A = sigma sqrt(4 CN0 / fs).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Sat:
    prn: int
    doppler: float  # Hz
    delay: float  # samples (0-based sample at which a primary-code period starts)
    phase: float  # rad
    cn0_dbhz: float = 45.0


def default_codegen():
    from . import native

    return native.gen_primary_code


def random_sats(rng, prns, spc, cn0_dbhz=45.0, max_doppler=4500.0):
    return [Sat(int(p), float(rng.uniform(-max_doppler, max_doppler)),
                float(rng.uniform(0, spc)), float(rng.uniform(0, 2 * np.pi)), cn0_dbhz)
            for p in prns]


def make_if(settings, sats, n_samples, seed=3550, sigma=20.0, codegen=None, chunk=1 << 22,
            out=None, iq_sign=0, clean=False, code_doppler=True, pilot61_secondary=False):
    """Return int8[n_samples] of real IF samples (fileType 1), or -- iq_sign = +1 / -1 --
    int8[2*n_samples] of interleaved I/Q pairs (fileType 2) holding the analytic signal
    a(t) e^{+j th} (iq_sign +1) or its conjugate (iq_sign -1).  The reference mixes with
    exp(+j th) in both acquisition.m files and B2a/tracking.m (:309) and with exp(-j th) in
    B1C/{NB,WB}_tracking.m (:320 / :341), so a complex record needs iq_sign -1 for the former
    and +1 for the latter to correlate.
    clean=True (real records only): the float64 sum of the satellites' signals, no noise, not quantised -- the caller adds
    its own noise realisations (bench.cfg4_record builds a long record from several of them).
    code_doppler=False: the code runs at the nominal rate whatever the carrier Doppler, so that a block of whole code
    periods repeats seamlessly (with the Doppler-scaled rate the code phase of such a record has a sawtooth of
    f_d / carrFreqBasis x chips per block -- 0.02 chip at 1.5 kHz, a quarter of the BOC(6,1) correlation peak's half width).
    pilot61_secondary=True: the BOC(6,1) part of the B1C pilot carries the secondary-code chip S like its BOC(1,1) part (as
    the two sub-carriers of one pilot code do on the air).  SURVEY.md section 8d's model -- the default here -- puts S on the
    BOC(1,1) part only; WB_tracking's QMBOC prompt -sqrt(4/33) p61 + sqrt(29/33) p11 then changes magnitude with S
    (measured: 0.8e11 vs 1.5e11 in |P|^2), which the moments-based C/N0 estimator reads as noise."""
    codegen = codegen or default_codegen()
    rng = np.random.default_rng(seed)
    fs = float(settings.samplingFreq)
    fc = float(settings.codeFreqBasis)
    ncode = int(settings.codeLength)
    b1c = str(settings.signal).upper() == "B1C"
    sig = "B1C" if b1c else "B2A"
    prim = {s.prn: (np.asarray(codegen(sig, "data", s.prn), dtype=np.float32),
                    np.asarray(codegen(sig, "pilot", s.prn), dtype=np.float32)) for s in sats}
    n_periods = int(np.ceil(n_samples * fc / fs / ncode)) + 3
    syms = {s.prn: (rng.choice([-1.0, 1.0], n_periods).astype(np.float32),
                    rng.choice([-1.0, 1.0], n_periods).astype(np.float32)) for s in sats}
    if clean:
        assert not iq_sign and out is None
        out = np.empty(n_samples, dtype=np.float64)
    elif out is None:
        out = np.empty(n_samples * (2 if iq_sign else 1), dtype=np.int8)
    for a in range(0, n_samples, chunk):
        b = min(n_samples, a + chunk)
        n = np.arange(a, b, dtype=np.float64)
        acc = rng.normal(0.0, sigma, b - a) if not clean else np.zeros(b - a)
        if iq_sign:
            acc = acc + 1j * rng.normal(0.0, sigma, b - a)
        for s in sats:
            amp = sigma * np.sqrt(4.0 * 10 ** (s.cn0_dbhz / 10) / fs)
            fcode = fc * (1.0 + s.doppler / float(settings.carrFreqBasis)) if code_doppler else fc
            chips = (n - s.delay) * (fcode / fs)  # code phase in chips (may be < 0)
            period = np.floor(chips / ncode)
            cph = chips - period * ncode
            ci = np.minimum(cph.astype(np.int64), ncode - 1)
            pidx = (period.astype(np.int64) + 1) % n_periods
            d_sym, p_sym = syms[s.prn][0][pidx], syms[s.prn][1][pidx]
            cd, cp = prim[s.prn][0][ci], prim[s.prn][1][ci]
            th = 2 * np.pi * np.fmod((float(settings.IF) + s.doppler) * n / fs, 1.0) + s.phase
            if b1c:
                sub2 = np.floor(cph * 2).astype(np.int64) & 1  # 0 -> -c, 1 -> +c
                boc11 = (2.0 * sub2 - 1.0)
                sub12 = np.floor(cph * 12).astype(np.int64) % 12  # ii-1 -> (-1)^ii
                boc61 = np.where(sub12 % 2 == 0, -1.0, 1.0)
                s_i = 0.5 * d_sym * cd * boc11 - np.sqrt(1 / 11) * cp * boc61 * (p_sym if pilot61_secondary else 1.0)
                s_q = np.sqrt(29 / 44) * cp * boc11 * p_sym
                base = s_i + 1j * s_q
            else:  # d sin(th) + p cos(th) = Re[(p - j d) e^{j th}]
                base = p_sym * cp - 1j * (d_sym * cd)
            z = amp * base * np.exp(1j * th)
            if iq_sign:
                acc += z if iq_sign > 0 else np.conj(z)
            else:
                acc += z.real
        if iq_sign:
            out[2 * a:2 * b:2] = np.clip(np.rint(acc.real), -127, 127).astype(np.int8)
            out[2 * a + 1:2 * b:2] = np.clip(np.rint(acc.imag), -127, 127).astype(np.int8)
        elif clean:
            out[a:b] = acc
        else:
            out[a:b] = np.clip(np.rint(acc), -127, 127).astype(np.int8)
    return out
