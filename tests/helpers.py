"""Shared builders for the parity tests: seeded synthetic IF blocks in the BASELINE.json shapes."""
import numpy as np

import bds_amd
from bds_amd import synth


def spc_of(s):
    return int(np.floor(s.samplingFreq / (s.codeFreqBasis / s.codeLength) + 0.5))


def cfg1_b2a():
    """BASELINE.json configs[0]: B2a, 1 PRN, 3 Doppler bins, 10 ms @ 99.375 MS/s."""
    s = bds_amd.init_settings_b2a(acqSatelliteList=[19], acqSearchBand=400, acqStep=400, fineNoncoh=7)
    spc = spc_of(s)
    sats = [synth.Sat(19, 310.0, 0.37 * spc, 1.1, 47.0), synth.Sat(20, -200.0, 0.71 * spc, 0.3, 45.0)]
    x = synth.make_if(s, sats, 10 * spc, seed=3550)
    return s, x, sats


def small_b1c(prns=(3, 7, 12), n_codes=5, band=500):
    """B1C at a reduced sampling rate so the float64 oracle finishes in seconds."""
    s = bds_amd.init_settings_b1c(samplingFreq=12.5e6, IF=3.5e6, acqSatelliteList=list(prns), acqSearchBand=band)
    spc = spc_of(s)
    sats = [synth.Sat(3, 230.0, 40000.3, 1.0, 45.0), synth.Sat(12, -410.0, 99000.8, 2.0, 43.0)]
    x = synth.make_if(s, sats, n_codes * spc, seed=11)
    return s, x, sats


def medium_b2a(prns=(5, 9, 19, 33)):
    s = bds_amd.init_settings_b2a(acqSatelliteList=list(prns))
    spc = spc_of(s)
    rng = np.random.default_rng(5)
    sats = synth.random_sats(rng, [9, 19], spc, cn0_dbhz=46.0)
    x = synth.make_if(s, sats, 17 * spc, seed=6)
    return s, x, sats
