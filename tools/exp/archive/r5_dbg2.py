"""debug: the forced 768 x L2 plan cases of tests/test_acq_gpu.py::test_every_specialised_plan_pair -- where does carrFreq differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ.setdefault("BDS_LIB_PATH", os.path.join(os.path.dirname(__file__), "..", "..", "bds-3-b1c-b2a-sdr-receiver_amd", "libbds_mi355x_hooks.so"))
import bds_amd
from bds_amd import synth
from oracle import acquisition as oacq
from helpers import spc_of
ctx = bds_amd.get_context(0)
l1 = 768
for k, l2 in enumerate((1280, 2048, 3072, 4096)):
    os.environ["BDS_ACQ_FORCE_L1L2"] = f"{l1}x{l2}"
    ctx.reload_tuning()
    fs = 10.0e6 + 1000.0 * (l1 + k)
    if k % 2 == 0:
        s = bds_amd.init_settings_b2a(samplingFreq=fs, IF=2.5e6, acqSatelliteList=[7, 19, 33], acqSearchBand=600, acqStep=200, fineNoncoh=4)
        fn, n_codes = oacq.acquisition_b2a, 7
    else:
        s = bds_amd.init_settings_b1c(samplingFreq=fs, IF=2.5e6, acqSatelliteList=[7, 19, 33], acqSearchBand=200, acqStep=100)
        fn, n_codes = oacq.acquisition_b1c, 3
    spc = spc_of(s)
    sats = [synth.Sat(19, 130.0, 0.41 * spc, 0.7, 47.0), synth.Sat(33, -90.0, 0.83 * spc, 2.2, 45.0)]
    x = synth.make_if(s, sats, n_codes * spc, seed=500 + l1 + k)
    diag = {}
    ref = fn(x.astype(np.float64), s, diag)
    got = bds_amd.acquisition(x, s, verbose=False)
    print(k, l2, s.signal, "carr got", got.carrFreq[[6, 18, 32]], "ref", ref.carrFreq[[6, 18, 32]], "cp", got.codePhase[[6, 18, 32]], ref.codePhase[[6, 18, 32]])
    for p in (7, 19, 33):
        if got.carrFreq[p - 1] != ref.carrFreq[p - 1]:
            fb = oacq.freq_bins(s)
            fbin = diag[p]["fbin"] - 1
            b1c = s.signal == "B1C"
            nfine = int(round(s.acqStep / 25)) * (2 if b1c else 1) + 1
            freqs = fb[fbin] - (s.acqStep if b1c else s.acqStep / 2) + 25.0 * np.arange(nfine)
            cp = int(ref.codePhase[p - 1])
            m = ctx.acq_coherent_sums(s, p, cp, freqs, 1)
            g = ctx.acq_coherent_sums(s, p, cp, freqs, 2)
            print("  PRN", p, "oracle fine", diag[p]["fine"])
            if b1c and m.shape[0] == 2:
                print("  multi  ", (np.abs(m[0]) * 11 + np.abs(m[1]) * 29) / 40)
                print("  single ", (np.abs(g[0]) * 11 + np.abs(g[1]) * 29) / 40)
            else:
                print("  multi  ", np.abs(m).sum(axis=0))
                print("  single ", np.abs(g).sum(axis=0))
