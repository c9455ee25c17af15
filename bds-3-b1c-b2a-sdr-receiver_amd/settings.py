"""Receiver settings -- host-side mirror of the reference's ``initSettings()``.

Field names, meanings and defaults follow
  B1C: BDS-3_B1C/initSettings.m:48-151
  B2a: BDS-3_B2a/initSettings.m:44-130
(the reference keeps one ``initSettings.m`` per receiver directory; here the
extra field ``signal`` ('B1C' | 'B2A') says which receiver the struct belongs
to).  Only the fields the acquisition/tracking path reads are kept
(SURVEY.md Appendix D); PVT / plot settings are out of scope.
"""
from __future__ import annotations

from types import SimpleNamespace


class Settings(SimpleNamespace):
    """Flat settings struct (MATLAB struct <-> attribute access)."""

    def copy(self, **changes) -> "Settings":
        d = dict(self.__dict__)
        d.update(changes)
        return Settings(**d)


def init_settings_b1c(**overrides) -> Settings:
    """BDS-3_B1C/initSettings.m defaults (53 MS/s NUT4NT recording)."""
    acq_coh_t = overrides.get("acqCohT", 10)
    s = Settings(
        signal="B1C",
        fileName="Set_Jan17_2018_13_53_for_Jimi_ch0.bin",
        dataType="schar",
        fileType=1,
        IF=1590e6 - 1575.42e6,
        samplingFreq=53e6,
        FEBW=27e6,
        msToProcess=37000,
        acqSatelliteList=[19, 20],
        pilotACQflag=1,
        gpuACQflag=1,
        pilotTRKflag=2,
        numberOfChannels=10,
        skipNumberOfBytes=0,
        codeLength=10230,
        codeFreqBasis=1.023e6,
        carrFreqBasis=1575.42e6,
        skipAcquisition=0,
        acqSearchBand=5000,
        acqCohT=acq_coh_t,
        acqStep=1000 / acq_coh_t / 2,
        acqThreshold=7.5,
        resamplingThreshold=15e6,
        resamplingflag=0,
        dllDampingRatio=0.7,
        dllNoiseBandwidth=1,
        dllCorrelatorSpacing=0.06,
        pllDampingRatio=0.7,
        pllNoiseBandwidth=12,
        intTime=0.01,
        CNoInterval=50,
    )
    s.__dict__.update(overrides)
    return s


def init_settings_b2a(**overrides) -> Settings:
    """BDS-3_B2a/initSettings.m defaults (99.375 MS/s recording)."""
    s = Settings(
        signal="B2A",
        msToProcess=49000,
        numberOfChannels=12,
        skipNumberOfBytes=0,
        fileName="Beidou_B2a_IF_signal.bin",
        dataType="schar",
        fileType=1,
        IF=13.55e6,
        samplingFreq=99.375e6,
        codeLength=10230,
        codeFreqBasis=10.23e6,
        skipAcquisition=0,
        acqSatelliteList=[19, 20],
        acqSearchBand=5000,
        acqThreshold=1.5,
        acqStep=400,
        fineNoncoh=15,
        resamplingThreshold=50e6,
        resamplingflag=0,
        dllDampingRatio=0.7,
        dllNoiseBandwidth=2,
        dllCorrelatorSpacing=0.5,
        pllDampingRatio=0.7,
        pllNoiseBandwidth=20,
        intTime=0.001,
        pilotTRKflag=1,
        CNoInterval=200,
        carrFreqBasis=1176.45e6,
    )
    s.__dict__.update(overrides)
    return s
