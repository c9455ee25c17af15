#!/usr/bin/env python3
"""Tracking timing (BASELINE.json configs[3] shape, shortened): B1C wide-band tracking, 12 channels,
fs = 99.375 MS/s, N 10-ms epochs on a synthetic int8 record resident in HBM.
    python tools/bench_track.py [--epochs 100] [--mode WB|NB|B2A]
Prints ms/epoch, samples/s and GB/s on the int8 read.  (Noise-only record: the cost per epoch does not
depend on lock.)"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bds_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=100)
ap.add_argument("--mode", default="WB")
ap.add_argument("--channels", type=int, default=12)
a = ap.parse_args()
if a.mode == "B2A":
    s = bds_amd.init_settings_b2a(msToProcess=a.epochs, numberOfChannels=a.channels)
    spc = 99375
else:
    s = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, msToProcess=a.epochs * 10,
                                  numberOfChannels=a.channels, pilotTRKflag=2 if a.mode == "WB" else 1)
    spc = 993750
rng = np.random.default_rng(1)
n = (a.epochs + 2) * spc
base = min(n, 202 * spc)  # long records repeat a 202-epoch noise block (timing does not depend on the data)
x = np.clip(np.rint(rng.normal(0, 20, base)), -127, 127).astype(np.int8)
if base < n:
    x = np.tile(x, n // base + 1)[:n]
ch = [SimpleNamespace(PRN=p, acquiredFreq=s.IF + 100.0 * i, codePhase=float(1000 * i + 1), codeFreq=s.codeFreqBasis, status="T")
      for i, p in enumerate(range(1, a.channels + 1))]
ctx = bds_amd.get_context(0)
bds_amd.tracking(x, ch, s, mode=a.mode)  # warm-up (includes H2D)
t0 = time.perf_counter()
res, _ = bds_amd.tracking(x, ch, s, mode=a.mode)
wall = time.perf_counter() - t0
dev_ms = ctx.timing()["total_ms"]
samples = sum(np.diff(r.absoluteSample).sum() + spc for r in res)
print(json.dumps({"mode": a.mode, "channels": a.channels, "epochs": a.epochs, "device_ms": dev_ms,
                  "ms_per_epoch": dev_ms / a.epochs, "wall_s_incl_h2d": wall,
                  "Msamples_per_s": samples / dev_ms / 1e3, "int8_read_GBps": samples / dev_ms / 1e6,
                  "completed": [r.completed for r in res]}))
