"""round 5: where does channel 6 of the cfg4 record leave the oracle's trajectory (epoch 1531 in every strict carrier mode)?
Tracks that one channel over 1 540 epochs on the GPU and with the oracle (sample loops in C) and prints the per-epoch differences
around the event.   BDS_TRK_PREC=4 python tools/exp/r5_cfg4_flip.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bds_amd  # noqa: E402
import bench  # noqa: E402
from oracle import cfast  # noqa: E402

cfast.build()
CH = int(os.environ.get("CH", "6"))
N = int(os.environ.get("EPOCHS", "1540"))
base = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1)
s, ch, blocks, order, shift, n, spc = bench.cfg4_record(base, N)
x = bench.record_bytes(blocks, order, shift, n)
sub = [ch[CH]]
s1 = s.copy(numberOfChannels=1)
got, _ = bds_amd.tracking(x, sub, s1, mode="WB")
ref = cfast.tracking_parallel(x, sub, s1, mode="WB")
g, r = got[0], ref[0]
p = np.hypot(r.I_P, r.Q_P).max()
fields = ("I_E", "I_P", "I_L", "Q_E", "Q_P", "Q_L", "Pilot_I_E", "Pilot_I_P", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_P", "Pilot_Q_L")
d = np.stack([np.abs(getattr(g, f) - getattr(r, f)) for f in fields])
first = int(np.argmax((d.max(axis=0) > 1e-9 * p) | (np.abs(g.codeFreq - r.codeFreq) > 1e-9)))
print("PRN", r.PRN, "|P|max", p, "first epoch (0-based) with a difference > 1e-9 |P|:", first, "absoluteSample equal:", np.array_equal(g.absoluteSample, r.absoluteSample))
for k in range(max(0, first - 3), min(N, first + 6)):
    print(k, "dcodeFreq %.3e dcarrFreq %.3e remCode %.17g / %.17g" % (g.codeFreq[k] - r.codeFreq[k], g.carrFreq[k] - r.carrFreq[k], g.remCodePhase[k], r.remCodePhase[k]),
          " ".join("%s %.4g" % (f, getattr(g, f)[k] - getattr(r, f)[k]) for f in fields if abs(getattr(g, f)[k] - getattr(r, f)[k]) > 1e-9 * p))
dr = g.remCodePhase - r.remCodePhase
ks = np.nonzero(np.diff(np.concatenate([[0.0], dr])) != 0)[0]
print("epochs (0-based) where the remCodePhase difference CHANGES:", ks[:20], "values", dr[ks[:20]])
for k in ks[:4]:
    kk = max(0, k - 1)
    print("  before epoch", k, ": epoch", kk, "codeFreq %.17g / %.17g rem %.17g / %.17g blk %d ; then rem %.17g / %.17g" % (
        g.codeFreq[kk], r.codeFreq[kk], g.remCodePhase[kk], r.remCodePhase[kk], int(r.absoluteSample[k] - r.absoluteSample[kk]) if k > 0 else 0, g.remCodePhase[k], r.remCodePhase[k]))
k = first
print("state at the event: codeFreq %.17g remCode %.17g carrFreq %.17g remCarr %.17g absSample %d" % (r.codeFreq[k], r.remCodePhase[k], r.carrFreq[k], r.remCarrPhase[k], int(r.absoluteSample[k])))
np.savez(os.path.join(ROOT, "gpurun_out", "r05_cfg4_flip_state.npz"), codeFreq=r.codeFreq[k], rem=r.remCodePhase[k], carrFreq=r.carrFreq[k], remCarr=r.remCarrPhase[k],
         pos=r.absoluteSample[k], gpu=np.array([getattr(g, f)[k] for f in fields]), ref=np.array([getattr(r, f)[k] for f in fields]))
