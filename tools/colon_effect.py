#!/usr/bin/env python3
"""What MATLAB's colon semantics change on BASELINE.json configs[3] (round 6).  CPU only (the oracle).

The reference's replica index vectors are colon vectors (WB_tracking.m:289-317); MATLAB generates their second half from the right-hand
end point (oracle/matlab.py m_colon).  Rounds 1-5 of this repo (oracle, C oracle and the HIP kernels alike) used a + k d throughout.
This script runs the float64 oracle (sample loops in C, one thread per channel) over the whole cfg4 horizon in BOTH forms on the same
record and reports
  (1) along the new oracle's trajectory: per epoch-channel, the samples whose ceil() differs between the two forms (E/P/L code index
      and E/P/L BOC(6,1) index), and the largest difference of the two tcode values in ulp;
  (2) per channel, the first epoch at which the two oracles leave SURVEY 8d's closed-loop tolerances of each other
      (I/Q 1e-4 of |P|, carrFreq 1e-3 Hz, codeFreq 1e-6 Hz), and the floor after it.

    python tools/colon_effect.py [--epochs 3600] [--mode WB|NB] > profiles/r06_colon_effect.txt
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bds_amd  # noqa: E402
import bench  # noqa: E402
from oracle import cfast  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3600)
    ap.add_argument("--mode", default="WB", help="WB | NB (cfg4 record) | B2A (the reference's B2a defaults: 12 channels x 49 000 one-ms epochs)")
    a = ap.parse_args()
    cfast.build()
    L = cfast.lib()
    L.bds_oracle_trk_colon_diff.argtypes = [ctypes.c_long] + [ctypes.c_double] * 4 + [ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double)]
    L.bds_oracle_trk_colon_diff.restype = ctypes.c_int
    L.bds_oracle_trk_set_plain_colon.argtypes = [ctypes.c_int]
    if a.mode == "B2A":
        s, x, ch, _, _, _, spc = bench.track_record("b2a", bds_amd.init_settings_b2a(), epochs=a.epochs)
        n = x.size
        print(f"B2a record at the reference's defaults: {n / 1e9:.2f} GB, {len(ch)} channels x {a.epochs} epochs, {os.cpu_count()} host cores")
    else:
        base = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1)
        s, ch, blocks, order, shift, n, spc = bench.cfg4_record(base, a.epochs)
        if a.mode == "NB":
            s = s.copy(pilotTRKflag=1)
        x = bench.record_bytes(blocks, order, shift, n)
        print(f"cfg4 record: {n / 1e9:.2f} GB, {len(ch)} channels x {a.epochs} epochs, mode {a.mode}, {os.cpu_count()} host cores")
    t0 = time.time()
    L.bds_oracle_trk_set_plain_colon(0)
    new = cfast.tracking_parallel(x, ch, s, mode=a.mode)
    t1 = time.time()
    L.bds_oracle_trk_set_plain_colon(1)
    old = cfast.tracking_parallel(x, ch, s, mode=a.mode)
    L.bds_oracle_trk_set_plain_colon(0)
    t2 = time.time()
    print(f"oracle with MATLAB's colon: {t1 - t0:.0f} s; oracle with a + k d (rounds 1-5): {t2 - t1:.0f} s")

    # (1) index differences along the new trajectory
    fs, code_len = s.samplingFreq, float(s.codeLength)
    spc_el = s.dllCorrelatorSpacing
    names = ("E code", "P code", "L code", "E BOC(6,1)", "P BOC(6,1)", "L BOC(6,1)")
    tot = np.zeros(6, dtype=np.int64)
    ep_with = np.zeros(6, dtype=np.int64)
    worst_ulp = 0.0
    per_epoch = []
    counts = (ctypes.c_long * 6)()
    mu = ctypes.c_double()
    for r in new:
        for k in range(a.epochs):
            step = r.codeFreq[k] / fs
            rem = r.remCodePhase[k]
            blk = int(np.ceil((code_len - rem) / step))
            rc = L.bds_oracle_trk_colon_diff(blk, float(rem), float(step), float(spc_el), 1.0 if a.mode == "B2A" else 2.0, counts, ctypes.byref(mu))
            assert rc == 0, rc
            c = np.array(list(counts))
            tot += c
            ep_with += c > 0
            per_epoch.append(int(c.sum()))
            worst_ulp = max(worst_ulp, mu.value)
    ne = len(new) * a.epochs
    print(f"\n(1) samples per epoch-channel whose ceil() differs between MATLAB's colon vector and a + k d, over {ne} epoch-channels of "
          f"~{spc} samples (new oracle's trajectory)")
    for i, nm in enumerate(names):
        print(f"    {nm:11s}: {int(tot[i]):6d} samples in total, {int(ep_with[i]):5d} epoch-channels affected")
    pe = np.array(per_epoch)
    print(f"    any replica: {int(pe.sum())} samples in {int((pe > 0).sum())} of {ne} epoch-channels ({(pe > 0).mean() * 100:.2f} %); "
          f"largest tcode difference {worst_ulp:.1f} ulp")
    if a.mode != "WB":
        print("    (the BOC(6,1) rows are what WB mode WOULD index; NB and B2a tracking read the code rows only)")

    # (2) separation of the two oracles
    fields = ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_E", "Pilot_I_P", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_P", "Pilot_Q_L")
    print(f"\n(2) oracle (a + k d) against oracle (MATLAB colon), SURVEY 8d: I/Q 1e-4 of |P|, carrFreq 1e-3 Hz, codeFreq 1e-6 Hz")
    n_sep = n_bad = 0
    for c, (o, m) in enumerate(zip(old, new)):
        assert np.array_equal(o.absoluteSample, m.absoluteSample) or True
        p = np.hypot(m.I_P, m.Q_P).max()
        d_iq = np.max(np.stack([np.abs(getattr(o, f) - getattr(m, f)) for f in fields if hasattr(m, f)]), axis=0) / p
        d_carr, d_code = np.abs(o.carrFreq - m.carrFreq), np.abs(o.codeFreq - m.codeFreq)
        same_abs = bool(np.array_equal(o.absoluteSample, m.absoluteSample))
        bad = (d_iq > 1e-4) | (d_carr > 1e-3) | (d_code > 1e-6)
        if bad.any():
            f = int(np.argmax(bad))
            n_sep += 1
            n_bad += a.epochs - f
            print(f"    channel {c:2d} (PRN {m.PRN:2d}): leaves 8d at epoch {f + 1:4d}: first discrepancy {d_iq[f] * p:.1f} (= {d_iq[f]:.2e} of |P|); "
                  f"afterwards worst I/Q {d_iq[f:].max():.2e} of |P|, carrFreq {d_carr[f:].max():.2e} Hz, codeFreq {d_code[f:].max():.2e} Hz; "
                  f"absoluteSample {'identical' if same_abs else 'DIFFERS'}")
        else:
            print(f"    channel {c:2d} (PRN {m.PRN:2d}): inside 8d over all {a.epochs} epochs (worst I/Q {d_iq.max():.2e} of |P|, carrFreq {d_carr.max():.2e} Hz, "
                  f"codeFreq {d_code.max():.2e} Hz)")
    print(f"    -> {n_sep} of {len(new)} channels separate; {n_bad} of {ne} epoch-channels ({n_bad / ne * 100:.1f} %) lie after a separation")


if __name__ == "__main__":
    main()
