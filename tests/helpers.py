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


def as_complex(x_iq):
    """int8 I/Q pairs -> the complex row postProcessing.m:92-96 hands to acquisition."""
    return x_iq[0::2].astype(np.float64) + 1j * x_iq[1::2].astype(np.float64)


def cfg1_b2a_iq():
    """cfg1 with a fileType 2 record (interleaved I/Q int8, B2a/initSettings.m:61)."""
    s = bds_amd.init_settings_b2a(acqSatelliteList=[19, 20, 21], acqSearchBand=800, acqStep=400, fineNoncoh=7,
                                  fileType=2)
    spc = spc_of(s)
    sats = [synth.Sat(19, 310.0, 0.37 * spc, 1.1, 47.0), synth.Sat(20, -200.0, 0.71 * spc, 0.3, 45.0)]
    x = synth.make_if(s, sats, 10 * spc, seed=3551, iq_sign=-1)
    return s, x, sats


def small_b1c_iq(prns=(3, 7, 12), n_codes=5, band=500):
    s = bds_amd.init_settings_b1c(samplingFreq=12.5e6, IF=3.5e6, acqSatelliteList=list(prns), acqSearchBand=band,
                                  fileType=2)
    spc = spc_of(s)
    sats = [synth.Sat(3, 230.0, 40000.3, 1.0, 45.0), synth.Sat(12, -410.0, 99000.8, 2.0, 43.0)]
    x = synth.make_if(s, sats, n_codes * spc, seed=12, iq_sign=-1)
    return s, x, sats


def resample_b2a():
    """B2a with the reference's resampling pre-conditioner on (B2a/acquisition.m:54-124):
    fs = 99.375 MS/s > resamplingThreshold, fir1(700) + filtfilt, fs' = 48.06 MS/s."""
    s = bds_amd.init_settings_b2a(acqSatelliteList=[19, 20, 21], acqSearchBand=800, acqStep=400, resamplingflag=1)
    spc = spc_of(s)
    sats = [synth.Sat(19, 310.0, 0.37 * spc, 1.1, 47.0), synth.Sat(20, -200.0, 0.71 * spc, 0.3, 45.0)]
    x = synth.make_if(s, sats, 17 * spc, seed=3550)
    return s, x, sats


def resample_b1c(iq=False):
    """B1C, fs = 40 MS/s, IF = 10 MHz, resampling on: 9-MHz band-pass, fs' = 29 MS/s (B1C/acquisition.m:54-123)."""
    s = bds_amd.init_settings_b1c(samplingFreq=40e6, IF=10e6, acqSatelliteList=[3, 7, 12], acqSearchBand=500,
                                  resamplingflag=1, resamplingThreshold=15e6, fileType=2 if iq else 1)
    spc = spc_of(s)
    sats = [synth.Sat(3, 230.0, 120000.3, 1.0, 46.0), synth.Sat(12, -410.0, 299000.8, 2.0, 44.0)]
    x = synth.make_if(s, sats, 4 * spc, seed=13, iq_sign=-1 if iq else 0)
    return s, x, sats


def track_case(signal, mode, n_epochs, seed=21, iq=False, fs=None, IF=None):
    """Synthetic record + channels for the tracking tests at a reduced sampling rate.

    Returns (settings, file_bytes int8, channels) with channels filled the way preRun
    would from a perfect acquisition (codePhase = first sample of a code period)."""
    from types import SimpleNamespace

    if signal == "B2A":
        s = bds_amd.init_settings_b2a(samplingFreq=fs or 25e6, IF=IF or 6.5e6, msToProcess=n_epochs, numberOfChannels=3,
                                      CNoInterval=20, fileType=2 if iq else 1)
        sat_list = [synth.Sat(9, -1230.0, 12345.6, 2.0, 50.0), synth.Sat(19, 2210.0, 3001.2, 0.4, 47.0),
                    synth.Sat(33, 355.0, 20111.9, 1.3, 45.0)]
    else:
        flag = {"NB": 1, "WB": 2}[mode]
        s = bds_amd.init_settings_b1c(samplingFreq=fs or 12.5e6, IF=IF or 3.5e6, msToProcess=n_epochs * 10, numberOfChannels=3,
                                      pilotTRKflag=flag, CNoInterval=10, FEBW=10e6, fileType=2 if iq else 1)
        sat_list = [synth.Sat(3, 230.0, 40000.3, 1.0, 48.0), synth.Sat(12, -410.0, 99000.8, 2.0, 45.0),
                    synth.Sat(27, 1800.0, 7000.5, 0.2, 46.0)]
    spc = spc_of(s)
    # a complex record correlates under B2a/tracking.m's exp(+j th) when conjugated, under the B1C
    # trackers' exp(-j th) when not (synth.make_if)
    x = synth.make_if(s, sat_list, (n_epochs + 3) * spc, seed=seed,
                      iq_sign=(-1 if signal == "B2A" else 1) if iq else 0)
    chans = []
    for sat in sat_list:
        cf = s.IF + round(sat.doppler / 25) * 25
        code_freq = (s.codeFreqBasis - (cf - s.IF) / s.carrFreqBasis * s.codeFreqBasis) if signal == "B1C" else s.codeFreqBasis
        chans.append(SimpleNamespace(PRN=sat.prn, acquiredFreq=float(cf), codePhase=float(int(np.ceil(sat.delay)) + 1),
                                     codeFreq=float(code_freq), status="T"))
    return s, x, chans
