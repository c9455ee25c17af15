"""Code/carrier tracking loops (oracle; test infrastructure).

Restates, in float64,
  B2a/tracking.m:98-441          (1 ms epochs, data + pilot, +j carrier, i = imag)
  B1C/NB_tracking.m:107-448      (10 ms epochs, data + pilot BOC(1,1))
  B1C/WB_tracking.m:114-488      (+ pilot BOC(6,1), QMBOC composite)
  Common/calcLoopCoef.m:41-45, Common/calcLoopCoefCarr.m:41-56
  B1C/include/CalcWeighingFactor.m:43-81
  B1C/include/Calc_CNo_PLD.m:45-114, B2a/include/Calc_CNo_PLD.m:38-100
  B1C/include/preRun.m:44-76, B2a/include/preRun.m:44-76

The file handle of the reference (fid + fseek/ftell/fread) is modelled by
``RawFile`` over an int8 array of the raw file bytes.  The waitbar GUI
(B2a/tracking.m:126-130,201-222) has no counterpart.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
from scipy import integrate

from . import codes
from .matlab import m_colon, m_round, m_var


class RawFile:
    """fid stand-in: byte-addressed int8 stream with fseek('bof')/ftell/fread('schar')."""

    def __init__(self, data):
        self.data = data  # 1-D int8 array (np.memmap works)
        self.pos = 0

    def seek(self, offset_bytes: int):
        self.pos = int(offset_bytes)

    def tell(self) -> int:
        return self.pos

    def read(self, count: int):
        out = np.asarray(self.data[self.pos: self.pos + count], dtype=np.float64)
        self.pos += out.size
        return out, out.size

    def read_raw(self, count: int):
        """The same read without the conversion to double (for a ``correlate=`` evaluator that takes the int8 samples)."""
        out = np.ascontiguousarray(self.data[self.pos: self.pos + count], dtype=np.int8)
        self.pos += out.size
        return out, out.size


# --- loop coefficients ---------------------------------------------------------
def calc_loop_coef(lbw, zeta, k):
    """Common/calcLoopCoef.m:41-45."""
    wn = lbw * 8 * zeta / (4 * zeta ** 2 + 1)
    return k / (wn * wn), 2.0 * zeta / wn


def calc_loop_coef_carr(settings):
    """Common/calcLoopCoefCarr.m:41-56 -> (pf3, pf2, pf1)."""
    wn = 1.2 * settings.pllNoiseBandwidth
    t = settings.intTime
    return wn ** 3 * t ** 2, 2 * wn ** 2 * t, 2 * wn


def calc_weighing_factor(settings):
    """B1C/include/CalcWeighingFactor.m:43-81 (MATLAB integral() -> scipy quad,
    both adaptive Gauss-Kronrod to ~1e-10 relative)."""
    fc = settings.codeFreqBasis
    tc = 1 / fc
    br = settings.FEBW

    def g11(f):
        return tc * (np.sin(np.pi / 2 * f / fc) * np.sin(np.pi * f / fc) /
                     np.cos(np.pi / 2 * f / fc) * fc / f / np.pi) ** 2

    def g61(f):
        return tc * (np.sin(np.pi / 12 * f / fc) * np.sin(np.pi * f / fc) /
                     np.cos(np.pi / 12 * f / fc) * fc / f / np.pi) ** 2

    def gp(f):
        return 29 / 33 * g11(f) + 4 / 33 * g61(f)

    def integ(fun):
        # even integrands; split at 0 to stay off the removable singularity
        v, _ = integrate.quad(fun, 0, br / 2, limit=2000, epsabs=0, epsrel=1e-12)
        return 2 * v

    p11_2 = integ(lambda f: g11(f) * f ** 2)
    p11 = integ(g11)
    rem11 = (p11_2 / p11) ** 0.5
    pp_2 = integ(lambda f: gp(f) * f ** 2)
    pp = integ(gp)
    remp = (pp_2 / pp) ** 0.5
    t1 = 11 * p11 * rem11 ** 2
    t2 = 33 * pp * remp ** 2
    return t1 / (t1 + t2)


def calc_cno_pld(i_p, q_p, pil_i, pil_q, settings, pilot_mode):
    """*/include/Calc_CNo_PLD.m.  Inputs are the last CNoInterval prompt values.

    pilot_mode: 0 none; 1 = I/Q swapped (B2a, B1C NB: :84-87); 2 = as is (B1C WB: :80-83).
    Returns (CNo[3], PllDetector[2]).
    """
    cno = np.zeros(3)
    pld = np.zeros(2)
    t = settings.intTime

    def one(i, q):
        with np.errstate(all="ignore"):
            z = i ** 2 + q ** 2
            zm = np.mean(z)
            zv = m_var(z)
            pav = np.sqrt(zm ** 2 - zv)
            nv = 0.5 * (zm - pav)
            lin = np.abs((1 / t) * pav / (2 * nv))
            s = np.sum(i[i > 0]) - np.sum(i[i < 0])
            nbp = s ** 2 + np.sum(q) ** 2
            nbd = s ** 2 - np.sum(q) ** 2
            return lin, 10 * np.log10(lin), nbd / nbp

    d_lin, cno[0], pld[0] = one(np.asarray(i_p), np.asarray(q_p))
    p_lin = 0.0
    if pilot_mode == 2:
        p_lin, cno[1], pld[1] = one(np.asarray(pil_i), np.asarray(pil_q))
    elif pilot_mode == 1:
        p_lin, cno[1], pld[1] = one(np.asarray(pil_q), np.asarray(pil_i))
    with np.errstate(all="ignore"):
        cno[2] = 10 * np.log10(d_lin + p_lin)
    return cno, pld


# --- channel allocation ----------------------------------------------------------
def pre_run(acq, settings):
    """*/include/preRun.m:44-76.  B1C aids codeFreq with the acquired Doppler
    (B1C/include/preRun.m:71-73), B2a does not (B2a/include/preRun.m:70)."""
    nch = int(settings.numberOfChannels)
    ch = [SimpleNamespace(PRN=0, acquiredFreq=0.0, codePhase=0.0, codeFreq=0.0, status="-")
          for _ in range(nch)]
    order = np.argsort(-np.asarray(acq.peakMetric), kind="stable")  # sort(...,'descend')
    for ii in range(min(nch, int(np.sum(np.asarray(acq.carrFreq) != 0)))):
        p = int(order[ii])
        ch[ii].PRN = p + 1
        ch[ii].acquiredFreq = float(acq.carrFreq[p])
        ch[ii].codePhase = float(acq.codePhase[p])
        if str(settings.signal).upper() == "B1C":
            ch[ii].codeFreq = settings.codeFreqBasis - \
                (ch[ii].acquiredFreq - settings.IF) / settings.carrFreqBasis * settings.codeFreqBasis
        else:
            ch[ii].codeFreq = settings.codeFreqBasis
        ch[ii].status = "T"
    return ch


# --- tracking ---------------------------------------------------------------------
def _num_to_process(settings, mode):
    if mode == "B2A":
        return int(settings.msToProcess)  # B2a/tracking.m:100
    return int(m_round(settings.msToProcess / 1000 / settings.intTime))  # B1C/WB_tracking.m:56


def _template(n, m, mode, pilot):
    r = SimpleNamespace()
    r.status = "-"
    r.PRN = None
    r.absoluteSample = np.zeros(n)
    for f in ("codeFreq", "carrFreq", "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt",
              "remCodePhase", "remCarrPhase"):
        setattr(r, f, np.full(n, np.inf))
    for f in ("I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L"):
        setattr(r, f, np.zeros(n))
    if pilot:
        r.Pilot_I_P = np.zeros(n)
        r.Pilot_Q_P = np.zeros(n)
        if mode == "WB":
            for f in ("Pilot_I_E", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_L"):
                setattr(r, f, np.zeros(n))
    r.DataCNo = np.zeros(m)
    r.DataPLD = np.zeros(m)
    if pilot:
        r.PilotCNo = np.zeros(m)
        r.PilotPLD = np.zeros(m)
        setattr(r, "B2a_CNo" if mode == "B2A" else "B1C_CNo", np.zeros(m))
    return r


def tracking(fid: RawFile, channel, settings, mode=None, trace=None, correlate=None):
    """[trackResults, channel] = tracking(fid, channel, settings).

    mode: 'B2A' (B2a/tracking.m), 'NB' (B1C/NB_tracking.m), 'WB' (B1C/WB_tracking.m);
    default from settings.signal / settings.pilotTRKflag as B1C/postProcessing.m:137-143 does.
    correlate (optional): another evaluator of one epoch's sample loop -- ``correlate(raw_int8, blk, iq, rem_code, step, spc_el, scale,
    rem_carr, carr_freq, fs, b2a, dcode, pcode | None, p6code | None)`` returning ``(sums[18], t_p_last, trig_end)`` -- e.g. the C
    restatement (``oracle.cfast.trk_epoch``); the loop filters, discriminators and C/N0 stay this function's.
    """
    if mode is None:
        if str(settings.signal).upper() == "B2A":
            mode = "B2A"
        else:
            mode = "WB" if settings.pilotTRKflag == 2 else "NB"
    b2a = mode == "B2A"
    pilot = (settings.pilotTRKflag == 1) if mode in ("B2A", "NB") else (settings.pilotTRKflag == 2)
    n_proc = _num_to_process(settings, mode)
    cno_int = int(settings.CNoInterval)
    m = n_proc // cno_int
    nch = int(settings.numberOfChannels)
    res = [_template(n_proc, m, mode, pilot) for _ in range(nch)]

    spc_el = settings.dllCorrelatorSpacing  # earlyLateSpc
    code_len = int(settings.codeLength)
    pdi = settings.intTime
    tau1, tau2 = calc_loop_coef(settings.dllNoiseBandwidth, settings.dllDampingRatio, 1.0)
    pf3, pf2, pf1 = calc_loop_coef_carr(settings)
    factor = calc_weighing_factor(settings) if mode == "WB" else None
    adapt = 1 if settings.fileType == 1 else 2
    fs = settings.samplingFreq
    two_pi = 2.0 * np.pi
    s433, s2933 = np.sqrt(4 / 33), np.sqrt(29 / 33)

    for c in range(nch):
        ch = channel[c]
        if ch.PRN == 0:
            continue
        r = res[c]
        r.PRN = ch.PRN
        fid.seek(adapt * (int(settings.skipNumberOfBytes) + int(ch.codePhase) - 1))  # :151-153
        if b2a:
            d = codes.generate_b2a_data_code(ch.PRN, settings)
            dcode = np.concatenate([[d[code_len - 1]], d, [d[0]]])  # :158
            if pilot:
                p = codes.generate_b2a_pilot_code(ch.PRN, settings)
                pcode = np.concatenate([[p[code_len - 1]], p, [p[0]]])  # :164
        else:
            d = codes.generate_data_boc11(settings, ch.PRN)
            dcode = np.concatenate([[d[-1]], d, [d[0]]])  # WB:181
            if pilot:
                p = codes.generate_pilot_boc11(settings, ch.PRN)
                pcode = np.concatenate([[p[-1]], p, [p[0]]])  # WB:187
                if mode == "WB":
                    p6 = codes.generate_pilot_boc61(settings, ch.PRN)
                    p6code = np.concatenate([[p6[-1]], p6, [p6[0]]])  # WB:192

        code_freq = ch.codeFreq
        rem_code = 0.0
        carr_freq = ch.acquiredFreq
        carr_basis = ch.acquiredFreq
        rem_carr = 0.0
        old_nco = 0.0
        old_err = 0.0
        d2 = 0.0
        d1 = 0.0
        cno_val = np.zeros(3)
        tmp_cno = np.zeros(3)
        aborted = False

        for k in range(1, n_proc + 1):
            r.absoluteSample[k - 1] = fid.tell() / adapt  # :226
            step = code_freq / fs  # :230
            blk = int(np.ceil((code_len - rem_code) / step))  # :233
            if correlate is not None:
                raw8, nread = fid.read_raw(adapt * blk)
                if nread != adapt * blk:  # :250-254
                    aborted = True
                    break
                r.remCodePhase[k - 1] = rem_code  # :258
                scale = 1.0 if b2a else 2.0
                sums, t_p_last, trig_end = correlate(raw8, blk, adapt == 2, rem_code, step, spc_el, scale, rem_carr, carr_freq, fs, b2a,
                                                     dcode, pcode if pilot else None, p6code if (pilot and mode == "WB") else None)
                rem_code = (t_p_last + step) - code_len if b2a else t_p_last / 2 + step - code_len  # :295 / WB:327
                r.remCarrPhase[k - 1] = rem_carr  # :300
                rem_carr = float(np.fmod(trig_end, two_pi))  # :305
                I_E, Q_E, I_P, Q_P, I_L, Q_L = (float(v) for v in sums[0:6])
                if pilot:
                    pI_E, pQ_E, pI_P, pQ_P, pI_L, pQ_L = (float(v) for v in sums[6:12])
                    if mode == "WB":
                        sI_E, sQ_E, sI_P, sQ_P, sI_L, sQ_L = (float(v) for v in sums[12:18])
                        cI_E = -s433 * sI_E + s2933 * pQ_E  # WB:375-380 QMBOC composite
                        cQ_E = -s433 * sQ_E - s2933 * pI_E
                        cI_P = -s433 * sI_P + s2933 * pQ_P
                        cQ_P = -s433 * sQ_P - s2933 * pI_P
                        cI_L = -s433 * sI_L + s2933 * pQ_L
                        cQ_L = -s433 * sQ_L - s2933 * pI_L
            else:
                raw, nread = fid.read(adapt * blk)
                if adapt == 2:
                    raw = raw[0::2] + 1j * raw[1::2]
                if nread != adapt * blk:  # :250-254  partial results, status stays '-'
                    aborted = True
                    break
                r.remCodePhase[k - 1] = rem_code  # :258
                kk = np.arange(blk, dtype=np.float64)
                scale = 1.0 if b2a else 2.0

                def taps(off):
                    # (rem +- spc)[*2] : step[*2] : ((blksize-1)*step + rem +- spc)[*2]   (:260-263, NB:271-273, WB:289-292) --
                    # a MATLAB colon vector: second half generated from the right-hand end point (oracle/matlab.py m_colon)
                    t = m_colon((rem_code + off) * scale, step * scale, (((blk - 1) * step + rem_code) + off) * scale)
                    if len(t) != blk:  # MATLAB itself would stop at "tcode(blksize)" / the element-wise products
                        raise RuntimeError(f"colon vector has {len(t)} elements for blksize {blk}")
                    return t, np.ceil(t).astype(np.int64) + 1

                t_e, i_e = taps(-spc_el)
                t_l, i_l = taps(+spc_el)
                t_p, i_p = taps(0.0)
                e_c, l_c, p_c = dcode[i_e - 1], dcode[i_l - 1], dcode[i_p - 1]
                if pilot:
                    pe_c, pl_c, pp_c = pcode[i_e - 1], pcode[i_l - 1], pcode[i_p - 1]
                    if mode == "WB":  # WB:298,311,324
                        p6e = p6code[np.ceil(t_e * 6).astype(np.int64)]
                        p6l = p6code[np.ceil(t_l * 6).astype(np.int64)]
                        p6p = p6code[np.ceil(t_p * 6).astype(np.int64)]
                if b2a:
                    rem_code = (t_p[blk - 1] + step) - code_len  # :295
                else:
                    rem_code = t_p[blk - 1] / 2 + step - code_len  # WB:327

                r.remCarrPhase[k - 1] = rem_carr  # :300
                time = np.arange(blk + 1, dtype=np.float64) / fs  # :303
                trig = ((carr_freq * 2.0 * np.pi) * time) + rem_carr  # :304
                rem_carr = float(np.fmod(trig[blk], two_pi))  # :305
                if b2a:
                    carr = np.exp(1j * trig[:blk])  # :309
                    mixed = carr * raw
                    q_bb = mixed.real  # :313
                    i_bb = mixed.imag  # :314
                else:
                    carr = np.exp(-1j * trig[:blk])  # NB:320
                    mixed = carr * raw
                    i_bb = mixed.real
                    q_bb = mixed.imag

                I_E, Q_E = np.sum(e_c * i_bb), np.sum(e_c * q_bb)
                I_P, Q_P = np.sum(p_c * i_bb), np.sum(p_c * q_bb)
                I_L, Q_L = np.sum(l_c * i_bb), np.sum(l_c * q_bb)
                if pilot:
                    pI_E, pQ_E = np.sum(pe_c * i_bb), np.sum(pe_c * q_bb)
                    pI_P, pQ_P = np.sum(pp_c * i_bb), np.sum(pp_c * q_bb)
                    pI_L, pQ_L = np.sum(pl_c * i_bb), np.sum(pl_c * q_bb)
                    if mode == "WB":
                        sI_E, sQ_E = np.sum(p6e * i_bb), np.sum(p6e * q_bb)
                        sI_P, sQ_P = np.sum(p6p * i_bb), np.sum(p6p * q_bb)
                        sI_L, sQ_L = np.sum(p6l * i_bb), np.sum(p6l * q_bb)
                        # WB:375-380 QMBOC composite
                        cI_E = -s433 * sI_E + s2933 * pQ_E
                        cQ_E = -s433 * sQ_E - s2933 * pI_E
                        cI_P = -s433 * sI_P + s2933 * pQ_P
                        cQ_P = -s433 * sQ_P - s2933 * pI_P
                        cI_L = -s433 * sI_L + s2933 * pQ_L
                        cQ_L = -s433 * sQ_L - s2933 * pI_L

            with np.errstate(all="ignore"):
                carr_err = np.arctan(np.float64(Q_P) / np.float64(I_P)) / two_pi  # :337
                if pilot:
                    if b2a:
                        qi = (pI_P + 1j * pQ_P) * np.exp(-1j * np.pi / 2)  # :345
                        cq = np.arctan(np.float64(qi.imag) / np.float64(qi.real)) / two_pi  # :348
                        carr_err = (carr_err + cq) / 2  # :352
                    elif mode == "NB":
                        cq = np.arctan(np.float64(-pI_P) / np.float64(pQ_P)) / two_pi  # NB:357
                        carr_err = (carr_err * 11 + cq * 29) / 40  # NB:360
                    else:
                        cq = np.arctan(np.float64(cQ_P) / np.float64(cI_P)) / two_pi  # WB:392
                        carr_err = (carr_err * 1 + cq * 3) / 4  # WB:395
                d2 = d2 + carr_err * pf3  # :356
                d1 = d2 + carr_err * pf2 + d1  # :357
                carr_nco = d1 + carr_err * pf1  # :358
                r.carrFreq[k - 1] = carr_freq  # :361
                carr_freq = carr_basis + carr_nco  # :363

                def env(a, b):
                    return np.sqrt(a * a + b * b)

                e_, l_ = env(I_E, Q_E), env(I_L, Q_L)
                code_err = (e_ - l_) / (e_ + l_)  # :366
                if not b2a:
                    code_err = code_err * (1 - spc_el)  # WB:409-410
                if pilot:
                    if mode == "WB":
                        pe_, pl_ = env(cI_E, cQ_E), env(cI_L, cQ_L)
                    else:
                        pe_, pl_ = env(pI_E, pQ_E), env(pI_L, pQ_L)
                    pce = (pe_ - pl_) / (pe_ + pl_)
                    if b2a:
                        code_err = (code_err + pce) / 2  # :377
                    elif mode == "NB":
                        code_err = (code_err * 11 + pce * (1 - spc_el) * 29) / 40  # NB:381-384
                    else:
                        code_err = code_err * factor + pce * (1 - spc_el) * (1 - factor)  # WB:418
                code_nco = old_nco + (tau2 / tau1) * (code_err - old_err) + code_err * (pdi / tau1)  # :381
                old_nco = code_nco
                old_err = code_err
                r.codeFreq[k - 1] = code_freq  # :387
                code_freq = ch.codeFreq - code_nco  # :389

            r.dllDiscr[k - 1] = code_err
            r.dllDiscrFilt[k - 1] = code_nco
            r.pllDiscr[k - 1] = carr_err
            r.pllDiscrFilt[k - 1] = carr_nco
            r.I_E[k - 1], r.I_P[k - 1], r.I_L[k - 1] = I_E, I_P, I_L
            r.Q_E[k - 1], r.Q_P[k - 1], r.Q_L[k - 1] = Q_E, Q_P, Q_L
            if pilot:
                if mode == "WB":
                    r.Pilot_I_E[k - 1], r.Pilot_Q_E[k - 1] = cI_E, cQ_E
                    r.Pilot_I_P[k - 1], r.Pilot_Q_P[k - 1] = cI_P, cQ_P
                    r.Pilot_I_L[k - 1], r.Pilot_Q_L[k - 1] = cI_L, cQ_L
                else:
                    r.Pilot_I_P[k - 1], r.Pilot_Q_P[k - 1] = pI_P, pQ_P
            if trace is not None:
                # NCO state this epoch was correlated with + the raw correlator sums
                # (order of include/bds_mi355x.h bds_track_correlate)
                raw18 = [I_E, Q_E, I_P, Q_P, I_L, Q_L]
                raw18 += [pI_E, pQ_E, pI_P, pQ_P, pI_L, pQ_L] if pilot else [0.0] * 6
                raw18 += [sI_E, sQ_E, sI_P, sQ_P, sI_L, sQ_L] if (pilot and mode == "WB") else [0.0] * 6
                trace.append(dict(ch=c, k=k, blk=blk, pos=r.absoluteSample[k - 1], rem=r.remCodePhase[k - 1],
                                  codeFreq=r.codeFreq[k - 1], remCarr=r.remCarrPhase[k - 1],
                                  carrFreq=r.carrFreq[k - 1], sums=np.array(raw18, dtype=np.float64)))

            if k % cno_int == 0:  # :411-433
                sl = slice(k - cno_int, k)
                pm = 0
                if pilot:
                    pm = 2 if mode == "WB" else 1
                cno_val, pll_det = calc_cno_pld(
                    r.I_P[sl], r.Q_P[sl],
                    r.Pilot_I_P[sl] if pilot else None, r.Pilot_Q_P[sl] if pilot else None,
                    settings, pm)
                cc = k // cno_int - 1
                r.DataCNo[cc] = cno_val[0] * 0.5 + tmp_cno[0] * 0.5
                r.DataPLD[cc] = pll_det[0]
                if pilot:
                    r.PilotCNo[cc] = cno_val[1] * 0.5 + tmp_cno[1] * 0.5
                    getattr(r, "B2a_CNo" if b2a else "B1C_CNo")[cc] = cno_val[2] * 0.5 + tmp_cno[2] * 0.5
                    r.PilotPLD[cc] = pll_det[1]
            tmp_cno = cno_val  # :434

        if aborted:
            # B2a/tracking.m:250-254: message, fclose(fid), return -- later channels untouched
            break
        r.status = ch.status  # :441
    return res, channel
