#!/usr/bin/env python3
"""Long-horizon closed-loop tracking fixtures (BASELINE.json configs[3] length): expected outputs from the float64
oracle over the FULL number of epochs the reference processes -- 3 600 ten-millisecond epochs of B1C wide-band
tracking (36 000 ms, 2 channels) and 49 000 one-millisecond epochs of B2a tracking (1 channel) -- so that drift of
the GPU loops against the oracle accumulates over the real horizon (SURVEY.md section 8d tolerances).

    python tests/golden/make_long_tracking.py        (a few minutes; writes trk_wb_long.npz, trk_b2a_long.npz)

A 36-s record is 450 MB of int8 noise and cannot be committed; the fixture holds ONE block of it (10 resp. 50 code
periods, ~1.25 MB) and the record is that block repeated.  The block is built to continue into itself: every
satellite's code period starts within its first sample (a fractional delay: with the code boundaries exactly ON
sample instants the loops would settle where ceil() of the code phase flips on rounding noise, a degeneracy no real
recording has; the whole block is then rotated by `shift` samples so that channels start mid-record), carrier frequencies sit on the grid fs / block_length (whole cycles per block), and
Dopplers are small enough for the code to slip only a few hundredths of a chip per block seam.  Both the oracle and the GPU read the same repeated record, so parity
does not depend on how well the seams match; matching them just keeps the loops locked like on a real recording.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bds_amd  # noqa: E402
from bds_amd import synth  # noqa: E402
from oracle import tracking as otrk  # noqa: E402

F64 = ("carrFreq", "codeFreq", "remCodePhase", "remCarrPhase", "absoluteSample")
F32 = ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_P", "Pilot_Q_P")


def record(block, shift, n_samples):
    b = np.roll(block, shift)
    reps = -(-n_samples // b.size)
    return np.tile(b, reps)[:n_samples]


def run(name, s, sats, block_len, shift, n_epochs, mode, seed):
    from types import SimpleNamespace

    block = synth.make_if(s, sats, block_len, seed=seed)
    spc = int(np.floor(s.samplingFreq / (s.codeFreqBasis / s.codeLength) + 0.5))
    x = record(block, shift, (n_epochs + 4) * spc + shift)
    chans = []
    for sat in sats:
        cf = s.IF + round(sat.doppler / 25) * 25  # what acquisition's 25-Hz fine grid would hand over
        code_freq = (s.codeFreqBasis - (cf - s.IF) / s.carrFreqBasis * s.codeFreqBasis) if mode != "B2A" else s.codeFreqBasis
        chans.append(SimpleNamespace(PRN=sat.prn, acquiredFreq=float(cf), codePhase=float(shift + int(np.ceil(sat.delay)) + 1),
                                     codeFreq=float(code_freq), status="T"))
    t0 = time.time()
    res, _ = otrk.tracking(otrk.RawFile(x), chans, s, mode=mode)
    print(f"{name}: oracle {time.time() - t0:.0f} s over {n_epochs} epochs x {len(chans)} channels; record {x.size / 1e6:.0f} MB")
    for r in res:
        lock = np.abs(r.I_P[n_epochs // 2:]).mean() / max(np.abs(r.Q_P[n_epochs // 2:]).mean(), 1e-9)
        print(f"   PRN {r.PRN}: status {r.status}, |I_P|/|Q_P| over the second half {lock:.1f}, carrFreq end {r.carrFreq[-1]:.3f}")
        assert r.status == "T" and lock > 3, "the loops must stay locked for the fixture to mean anything"
    out = {f: np.stack([getattr(r, f) for r in res]) for f in F64}
    out.update({f: np.stack([getattr(r, f) for r in res]).astype(np.float32) for f in F32})
    sig = "B2a_CNo" if mode == "B2A" else "B1C_CNo"
    out["SigCNo"] = np.stack([getattr(r, sig) for r in res])
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), block=block, shift=shift, n_epochs=n_epochs, mode=mode,
                        settings=json.dumps(s.__dict__), channels=json.dumps([c.__dict__ for c in chans]), **out)
    print(f"   wrote {name}.npz ({os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.1f} MB)")


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    # B1C wide-band: 12.5 MS/s, 10 code periods per block (1 250 000 samples), carriers on the 10-Hz grid
    n_ep = 3600
    s = bds_amd.init_settings_b1c(samplingFreq=12.5e6, IF=3.5e6, msToProcess=n_ep * 10, numberOfChannels=2, pilotTRKflag=2,
                                  CNoInterval=50, FEBW=10e6)
    sats = [synth.Sat(3, 230.0, 0.37, 1.0, 47.0), synth.Sat(12, -410.0, 0.61, 2.0, 45.0)]
    if only in ("", "wb"):
        run("trk_wb_long", s, sats, 1250000, 40000, n_ep, "WB", seed=91)
    if only not in ("", "b2a"):
        sys.exit(0)
    # B2a: 25 MS/s, 50 code periods per block (1 250 000 samples), carrier on the 20-Hz grid.  The Doppler is kept
    # small on purpose: B2a/preRun.m:70 starts every channel at the nominal code rate and tracking.m has no carrier
    # aiding, so its 2-Hz DLL has to absorb the code Doppler (10.23e6 / 1176.45e6 chips/s per Hz) on its own and
    # loses lock beyond a few hundred Hz -- in the reference exactly as in the oracle.  The code slips 0.04 chip per
    # block seam at -110 Hz, which the loop rides through.
    n_ep = 49000
    s = bds_amd.init_settings_b2a(samplingFreq=25e6, IF=6500010.0, msToProcess=n_ep, numberOfChannels=1, CNoInterval=200)
    run("trk_b2a_long", s, [synth.Sat(19, -110.0, 0.43, 0.7, 48.0)], 1250000, 9000, n_ep, "B2A", seed=92)
