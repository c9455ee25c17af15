"""ctypes binding of ``libbds_mi355x.so`` (the C ABI declared in include/bds_mi355x.h).

This is the repo's counterpart of the MEX gateway (mex/bds_mex.c): the same
entry points, called from Python instead of MATLAB.  There is no CPU fallback --
if the library is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

BDS_MAX_PRN = 63
SIGNAL = {"B1C": 1, "B2A": 2}
TRACK_MODE = {"B2A": 0, "NB": 1, "WB": 2}
CODE_KIND = {"data": 0, "pilot": 1, "data_boc11": 2, "pilot_boc11": 3, "pilot_boc61": 4, "pilot_secondary": 5}
CODE_LEN = {0: 10230, 1: 10230, 2: 20460, 3: 20460, 4: 122760, 5: 1800}

# BDS_LIB_PATH: load another build of the library -- libbds_mi355x_hooks.so (the test-hooks build: tests/conftest.py,
# tools/exp/), the debug build, timing variants; default = the in-tree RELEASE build
_LIB_PATH = os.environ.get("BDS_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbds_mi355x.so")


class BdsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbds_mi355x error {code}: {msg}")
        self.code = code


class Settings(C.Structure):
    _fields_ = [
        ("signal", C.c_int32), ("fileType", C.c_int32),
        ("samplingFreq", C.c_double), ("IF", C.c_double), ("codeFreqBasis", C.c_double),
        ("carrFreqBasis", C.c_double),
        ("codeLength", C.c_int32), ("numberOfChannels", C.c_int32),
        ("skipNumberOfBytes", C.c_int64), ("msToProcess", C.c_double),
        ("acqSearchBand", C.c_double), ("acqStep", C.c_double), ("acqThreshold", C.c_double),
        ("acqCohT", C.c_double), ("pilotACQflag", C.c_int32), ("fineNoncoh", C.c_int32),
        ("resamplingThreshold", C.c_double), ("resamplingflag", C.c_int32), ("n_acq", C.c_int32),
        ("acqSatelliteList", C.c_int32 * BDS_MAX_PRN),
        ("pilotTRKflag", C.c_int32), ("intTime", C.c_double),
        ("dllCorrelatorSpacing", C.c_double), ("dllDampingRatio", C.c_double),
        ("dllNoiseBandwidth", C.c_double), ("pllNoiseBandwidth", C.c_double),
        ("CNoInterval", C.c_int32), ("dataType", C.c_int32), ("FEBW", C.c_double),
    ]


class Channel(C.Structure):
    _fields_ = [("PRN", C.c_int32), ("status", C.c_int32), ("acquiredFreq", C.c_double),
                ("codePhase", C.c_double), ("codeFreq", C.c_double)]


_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)

TRACK_FIELDS = ["absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L",
                "Pilot_I_P", "Pilot_Q_P", "Pilot_I_E", "Pilot_I_L", "Pilot_Q_E", "Pilot_Q_L",
                "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt", "remCodePhase", "remCarrPhase",
                "DataCNo", "DataPLD", "PilotCNo", "PilotPLD", "SigCNo"]


class TrackOut(C.Structure):
    _fields_ = ([("n_ch", C.c_int32), ("n_epochs", C.c_int32), ("n_cno", C.c_int32), ("reserved0", C.c_int32)]
                + [(f, _DP) for f in TRACK_FIELDS] + [("completed", _IP), ("status", _IP)])


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("forward_ms", C.c_double), ("search_ms", C.c_double),
                ("refine_ms", C.c_double), ("cell_pair_ms", C.c_double), ("cells_per_pair", C.c_double),
                ("n_pairs", C.c_int64), ("fft_len", C.c_int64), ("n_circ", C.c_int64),
                ("n_bins", C.c_int32), ("n_prn", C.c_int32), ("n_comp", C.c_int32), ("half_storage", C.c_int32),
                ("rows_ms", C.c_double), ("cols_ms", C.c_double), ("n_extra", C.c_int64), ("shader_clock_GHz", C.c_double),
                ("plan_l1", C.c_int32), ("plan_l2", C.c_int32), ("rows_kernel", C.c_int32), ("cols_kernel", C.c_int32),
                ("kernel_flags", C.c_int32), ("refine_path", C.c_int32)]


class AcqJob(C.Structure):
    _fields_ = [("settings", C.POINTER(Settings)), ("samples", C.POINTER(C.c_int8)), ("n_samples", C.c_size_t),
                ("is_complex", C.c_int32), ("max_prn", C.c_int32), ("carrFreq", _DP), ("codePhase", _DP),
                ("peakMetric", _DP), ("detected", _IP)]


EXPORTS = [
    "bds_create", "bds_destroy", "bds_reload_tuning", "bds_last_error", "bds_device_name", "bds_abi_check", "bds_build_flags", "bds_gen_code", "bds_acquire",
    "bds_acq_load", "bds_acq_prepare", "bds_acq_run", "bds_acq_set_pair_budget_gb", "bds_resample_plan", "bds_fir1_bandpass", "bds_frame_sync", "bds_sync_pattern", "bds_unpack_cplx", "bds_unpack_cplx_file", "bds_acq_grid", "bds_acq_peaks", "bds_acq_candidates", "bds_acq_coherent_sums", "bds_get_timing",
    "bds_track", "bds_track_mem", "bds_track_loaded_bytes", "bds_track_correlate", "bds_calc_loop_coef", "bds_calc_loop_coef_carr",
    "bds_calc_weighing_factor", "bds_pre_run", "bds_pre_run_device", "bds_acquire_track",
    "bds_multi_create", "bds_multi_destroy", "bds_multi_last_error", "bds_multi_size", "bds_multi_ctx",
    "bds_multi_rccl_ranks", "bds_acquire_multi", "bds_shard_jobs", "bds_acq_job_cost",
]

_lib = None
HOOKS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbds_mi355x_hooks.so")


def has_test_hooks():
    """True when the loaded library is the test-hooks build (reads the tuning / test environment switches)."""
    return bool(lib().bds_build_flags() & 1)


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: run ./build.sh (or __graft_entry__.build()); "
                          "there is no CPU fallback")
    # Share ONE HIP runtime with PyTorch when both live in a process: torch bundles its own
    # libamdhip64.so.7; importing it first makes the loader reuse it for our DT_NEEDED entry.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing only; the library also runs without it
        pass
    L = C.CDLL(_LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    i8p = C.POINTER(C.c_int8)
    SP = C.POINTER(Settings)
    L.bds_create.restype, L.bds_create.argtypes = vp, [i32]
    L.bds_destroy.restype, L.bds_destroy.argtypes = None, [vp]
    L.bds_reload_tuning.restype, L.bds_reload_tuning.argtypes = i32, [vp]
    L.bds_build_flags.restype, L.bds_build_flags.argtypes = i32, []
    L.bds_track_loaded_bytes.restype, L.bds_track_loaded_bytes.argtypes = C.c_longlong, [vp]
    L.bds_multi_create.restype, L.bds_multi_create.argtypes = vp, [i32, _IP]
    L.bds_multi_destroy.restype, L.bds_multi_destroy.argtypes = None, [vp]
    L.bds_multi_last_error.restype, L.bds_multi_last_error.argtypes = C.c_char_p, [vp]
    L.bds_multi_size.restype, L.bds_multi_size.argtypes = i32, [vp]
    L.bds_multi_ctx.restype, L.bds_multi_ctx.argtypes = vp, [vp, i32]
    L.bds_multi_rccl_ranks.restype, L.bds_multi_rccl_ranks.argtypes = i32, [vp]
    L.bds_acquire_multi.restype, L.bds_acquire_multi.argtypes = i32, [vp, i32, C.POINTER(AcqJob)]
    L.bds_shard_jobs.restype, L.bds_shard_jobs.argtypes = i32, [i32, _DP, i32, _IP]
    L.bds_acq_job_cost.restype, L.bds_acq_job_cost.argtypes = C.c_double, [SP]
    L.bds_last_error.restype, L.bds_last_error.argtypes = C.c_char_p, [vp]
    L.bds_device_name.restype, L.bds_device_name.argtypes = i32, [vp, C.c_char_p, i32]
    L.bds_gen_code.restype, L.bds_gen_code.argtypes = i32, [i32, i32, i32, i8p, i32]
    L.bds_acquire.restype = i32
    L.bds_acquire.argtypes = [vp, SP, i8p, sz, i32, i32, _DP, _DP, _DP, _IP]
    L.bds_acq_load.restype, L.bds_acq_load.argtypes = i32, [vp, SP, i8p, sz, i32]
    L.bds_acq_prepare.restype, L.bds_acq_prepare.argtypes = i32, [vp, SP]
    L.bds_acq_run.restype = i32
    L.bds_acq_run.argtypes = [vp, SP, _IP, i32, i32, _DP, _DP, _DP, _IP]
    L.bds_acq_grid.restype, L.bds_acq_grid.argtypes = i32, [vp, C.POINTER(C.c_float), _IP, i32]
    L.bds_acq_candidates.restype, L.bds_acq_candidates.argtypes = i32, [vp, i32, _IP, C.POINTER(C.c_int64), i32]
    L.bds_acq_peaks.restype, L.bds_acq_peaks.argtypes = i32, [vp, i32, _DP, _DP, _IP]
    L.bds_acq_coherent_sums.restype = i32
    L.bds_acq_coherent_sums.argtypes = [vp, SP, i32, C.c_int64, _DP, i32, i32, _DP, i32]
    L.bds_get_timing.restype, L.bds_get_timing.argtypes = i32, [vp, C.POINTER(Timing)]
    L.bds_track.restype = i32
    L.bds_track.argtypes = [vp, SP, C.c_char_p, i32, C.POINTER(Channel), C.POINTER(TrackOut)]
    L.bds_track_mem.restype = i32
    L.bds_track_mem.argtypes = [vp, SP, i8p, sz, i32, C.POINTER(Channel), C.POINTER(TrackOut)]
    L.bds_track_correlate.restype = i32
    L.bds_track_correlate.argtypes = [vp, SP, i8p, sz, i32, _IP, _DP, _DP]
    L.bds_calc_loop_coef.restype = None
    L.bds_calc_loop_coef.argtypes = [C.c_double, C.c_double, C.c_double, _DP, _DP]
    L.bds_calc_loop_coef_carr.restype, L.bds_calc_loop_coef_carr.argtypes = None, [SP, _DP, _DP, _DP]
    L.bds_calc_weighing_factor.restype, L.bds_calc_weighing_factor.argtypes = C.c_double, [SP]
    L.bds_pre_run.restype = i32
    L.bds_frame_sync.restype = i32
    L.bds_frame_sync.argtypes = [vp, i32, i32, _IP, _DP, i32, _IP, _IP, _IP, i32]
    L.bds_sync_pattern.restype, L.bds_sync_pattern.argtypes = i32, [i32, i32, C.POINTER(C.c_int8), i32]
    L.bds_unpack_cplx.restype, L.bds_unpack_cplx.argtypes = i32, [vp, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int8)]
    L.bds_unpack_cplx_file.restype, L.bds_unpack_cplx_file.argtypes = i32, [vp, C.c_char_p, C.c_char_p]
    L.bds_resample_plan.restype, L.bds_resample_plan.argtypes = i32, [SP, _DP, _DP, _DP]
    L.bds_fir1_bandpass.restype, L.bds_fir1_bandpass.argtypes = i32, [i32, C.c_double, C.c_double, _DP]
    L.bds_pre_run.argtypes = [SP, i32, _DP, _DP, _DP, C.POINTER(Channel)]
    L.bds_pre_run_device.restype, L.bds_pre_run_device.argtypes = i32, [vp, SP, i32, _DP, _DP, _DP, C.POINTER(Channel)]
    L.bds_acquire_track.restype = i32
    L.bds_acquire_track.argtypes = [vp, SP, i8p, sz, i32, i32, _DP, _DP, _DP, _IP, C.c_char_p, C.POINTER(Channel), C.POINTER(TrackOut)]
    L.bds_abi_check.restype, L.bds_abi_check.argtypes = i32, [i32, i32, i32, i32]
    if L.bds_abi_check(C.sizeof(Settings), C.sizeof(Channel), C.sizeof(TrackOut), C.sizeof(Timing)) != 0:
        raise ImportError("ctypes struct layout does not match libbds_mi355x.so (include/bds_mi355x.h changed?)")
    _lib = L
    return L


def debug_failures() -> int:
    """Failed device-side bounds checks since the last call; 0 on the product build (only the debug build, BDS_DEBUG=1
    ./build.sh, exports the counter) or when the library was never loaded."""
    if _lib is None or not hasattr(_lib, "bds_debug_failures"):
        return 0
    _lib.bds_debug_failures.restype = C.c_uint
    return int(_lib.bds_debug_failures())


def pack_settings(s) -> Settings:
    """MATLAB-style settings struct -> bds_settings.  Every field of SURVEY.md Appendix D that the selected
    receiver's initSettings.m defines is REQUIRED (a missing or misspelt field is an error naming it, never a silent
    default) -- the same rule as mex/bds_mex.c:pack_settings."""
    cs = Settings()
    sig = str(getattr(s, "signal", "")).upper()
    if sig not in SIGNAL:
        raise ValueError("settings.signal must be 'B1C' or 'B2A'")
    cs.signal = SIGNAL[sig]

    def need(name):
        if not hasattr(s, name):
            raise AttributeError(f"settings.{name} is missing")
        return getattr(s, name)

    # the library rejects anything but int8 samples (BDS_ERR_UNSUPPORTED names the field)
    cs.dataType = 0 if str(need("dataType")) in ("schar", "int8") else 1
    cs.fileType = int(need("fileType"))
    cs.samplingFreq = float(need("samplingFreq"))
    cs.IF = float(need("IF"))
    cs.codeFreqBasis = float(need("codeFreqBasis"))
    cs.carrFreqBasis = float(need("carrFreqBasis"))
    cs.codeLength = int(need("codeLength"))
    cs.numberOfChannels = int(need("numberOfChannels"))
    cs.skipNumberOfBytes = int(need("skipNumberOfBytes"))
    cs.msToProcess = float(need("msToProcess"))
    cs.acqSearchBand = float(need("acqSearchBand"))
    cs.acqStep = float(need("acqStep"))
    cs.acqThreshold = float(need("acqThreshold"))
    cs.resamplingThreshold = float(need("resamplingThreshold"))
    cs.resamplingflag = int(need("resamplingflag"))
    if sig == "B1C":  # B1C/initSettings.m:60,70,102
        cs.acqCohT = float(need("acqCohT"))
        cs.pilotACQflag = int(need("pilotACQflag"))
        cs.FEBW = float(need("FEBW"))
        cs.fineNoncoh = 1
    else:  # B2a/initSettings.m:84
        cs.fineNoncoh = int(need("fineNoncoh"))
        cs.acqCohT = 10.0
        cs.pilotACQflag = 1
        cs.FEBW = 0.0
    sats = [int(p) for p in np.atleast_1d(need("acqSatelliteList"))]
    if len(sats) > BDS_MAX_PRN:
        raise ValueError("settings.acqSatelliteList longer than 63")
    cs.n_acq = len(sats)
    for i, p in enumerate(sats):
        cs.acqSatelliteList[i] = p
    cs.pilotTRKflag = int(need("pilotTRKflag"))
    cs.intTime = float(need("intTime"))
    cs.dllCorrelatorSpacing = float(need("dllCorrelatorSpacing"))
    cs.dllDampingRatio = float(need("dllDampingRatio"))
    cs.dllNoiseBandwidth = float(need("dllNoiseBandwidth"))
    cs.pllNoiseBandwidth = float(need("pllNoiseBandwidth"))
    cs.CNoInterval = int(need("CNoInterval"))
    return cs


def _i8(a):
    a = np.ascontiguousarray(a, dtype=np.int8)
    return a, a.ctypes.data_as(C.POINTER(C.c_int8))


def gen_code(signal: str, kind: str, prn: int) -> np.ndarray:
    """bds_gen_code -> int8 array of +-1 (host side; works without a GPU)."""
    k = CODE_KIND[kind]
    out = np.empty(CODE_LEN[k], dtype=np.int8)
    rc = lib().bds_gen_code(SIGNAL[signal.upper()], k, int(prn), out.ctypes.data_as(C.POINTER(C.c_int8)), out.size)
    if rc < 0:
        raise BdsError(rc, f"bds_gen_code({signal}, {kind}, {prn})")
    return out


def sync_pattern(signal: str, prn: int = 1) -> np.ndarray:
    """+-1 frame-sync pattern: B1C pilot secondary code of the PRN (1800) or the B2a preamble x NH (120)."""
    out = np.empty(1800, dtype=np.int8)
    rc = lib().bds_sync_pattern(SIGNAL[signal.upper()], int(prn), out.ctypes.data_as(C.POINTER(C.c_int8)), out.size)
    if rc < 0:
        raise BdsError(rc, f"bds_sync_pattern({signal}, {prn})")
    return out[:rc].copy()


def shard_jobs(costs, world: int) -> np.ndarray:
    """bds_shard_jobs: rank of every job (longest-processing-time rule); host side, works without a GPU."""
    c = np.ascontiguousarray(costs, dtype=np.float64)
    out = np.zeros(c.size, dtype=np.int32)
    rc = lib().bds_shard_jobs(int(c.size), c.ctypes.data_as(_DP), int(world), out.ctypes.data_as(_IP))
    if rc < 0:
        raise BdsError(rc, "bds_shard_jobs")
    return out


def acq_job_cost(settings) -> float:
    """bds_acq_job_cost: relative cost of searching one PRN with these settings."""
    cs = pack_settings(settings)
    return float(lib().bds_acq_job_cost(C.byref(cs)))


class MultiContext:
    """bds_multi: one host process driving several GPUs (device_ids None = every visible device)."""

    def __init__(self, device_ids=None):
        self._lib = lib()
        if device_ids is None:
            self._h = self._lib.bds_multi_create(0, None)
        else:
            ids = np.ascontiguousarray(device_ids, dtype=np.int32)
            self._h = self._lib.bds_multi_create(int(ids.size), ids.ctypes.data_as(_IP))
        if not self._h:
            raise BdsError(-2, self._lib.bds_multi_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bds_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self) -> int:
        return int(self._lib.bds_multi_size(self._h))

    def rccl_ranks(self) -> int:
        return int(self._lib.bds_multi_rccl_ranks(self._h))

    def acquire(self, jobs):
        """jobs: list of (settings, int8 samples, is_complex) -- one entry per signal.
        Returns a list of (carrFreq, codePhase, peakMetric, detected), one per signal."""
        n = len(jobs)
        arr = (AcqJob * n)()
        keep, outs = [], []
        for i, (settings, samples, is_complex) in enumerate(jobs):
            cs = pack_settings(settings)
            a, p = _i8(samples)
            max_prn = max(int(q) for q in np.atleast_1d(settings.acqSatelliteList))
            carr, cph, pm = np.zeros(max_prn), np.zeros(max_prn), np.zeros(max_prn)
            det = np.zeros(max_prn, dtype=np.int32)
            arr[i].settings = C.pointer(cs)
            arr[i].samples = p
            arr[i].n_samples = a.size // (2 if is_complex else 1)
            arr[i].is_complex = int(bool(is_complex))
            arr[i].max_prn = max_prn
            arr[i].carrFreq, arr[i].codePhase, arr[i].peakMetric = (v.ctypes.data_as(_DP) for v in (carr, cph, pm))
            arr[i].detected = det.ctypes.data_as(_IP)
            keep.append((cs, a))
            outs.append((carr, cph, pm, det))
        rc = self._lib.bds_acquire_multi(self._h, n, arr)
        if rc < 0:
            raise BdsError(rc, self._lib.bds_multi_last_error(self._h).decode())
        return outs


def gen_primary_code(signal: str, kind: str, prn: int) -> np.ndarray:
    return gen_code(signal, kind, prn)


class Context:
    """One bds_ctx (one GPU)."""

    def __init__(self, device: int = 0):
        self._lib = lib()
        self._h = self._lib.bds_create(int(device))
        if not self._h:
            raise BdsError(-2, self._lib.bds_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bds_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc < 0:
            raise BdsError(rc, self._lib.bds_last_error(self._h).decode())
        return rc

    def reload_tuning(self):
        """Re-read the BDS_* environment knobs into this context (they are read once at creation)."""
        self._check(self._lib.bds_reload_tuning(self._h))

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._check(self._lib.bds_device_name(self._h, buf, 256))
        return buf.value.decode()

    # -- acquisition -----------------------------------------------------------------
    def acq_load(self, settings, samples, is_complex=False):
        cs = pack_settings(settings)
        a, p = _i8(samples)
        n = a.size // 2 if is_complex else a.size
        self._check(self._lib.bds_acq_load(self._h, C.byref(cs), p, n, int(is_complex)))

    def acq_prepare(self, settings):
        cs = pack_settings(settings)
        self._check(self._lib.bds_acq_prepare(self._h, C.byref(cs)))

    def acq_set_pair_budget(self, gib):
        """bds_acq_set_pair_budget_gb: serving mode of the search -- several PRNs' Doppler rows per launch pair, inter-pass buffer of
        `gib` GiB ("auto" / negative: 60 % of the free device memory; 0: the minimal footprint, one PRN per pair; the library default is 40).  Takes effect at the next acq_run, in both directions (a larger buffer is given back)."""
        g = -1.0 if (isinstance(gib, str) and gib.lower().startswith("a")) else float(gib)
        self._lib.bds_acq_set_pair_budget_gb.argtypes = [C.c_void_p, C.c_double]
        self._check(self._lib.bds_acq_set_pair_budget_gb(self._h, g))

    def acq_run(self, settings, prn_list=None):
        cs = pack_settings(settings)
        max_prn = max(int(p) for p in np.atleast_1d(settings.acqSatelliteList))
        carr = np.zeros(max_prn)
        cph = np.zeros(max_prn)
        pm = np.zeros(max_prn)
        det = np.zeros(max_prn, dtype=np.int32)
        if prn_list is None:
            pl, npl = None, 0
        else:
            arr = np.ascontiguousarray(prn_list, dtype=np.int32)
            pl, npl = arr.ctypes.data_as(_IP), arr.size
        self._check(self._lib.bds_acq_run(self._h, C.byref(cs), pl, npl, max_prn,
                                          carr.ctypes.data_as(_DP), cph.ctypes.data_as(_DP),
                                          pm.ctypes.data_as(_DP), det.ctypes.data_as(_IP)))
        return carr, cph, pm, det

    def acquire(self, settings, samples, is_complex=False):
        cs = pack_settings(settings)
        a, p = _i8(samples)
        n = a.size // 2 if is_complex else a.size
        max_prn = max(int(q) for q in np.atleast_1d(settings.acqSatelliteList))
        carr = np.zeros(max_prn)
        cph = np.zeros(max_prn)
        pm = np.zeros(max_prn)
        det = np.zeros(max_prn, dtype=np.int32)
        self._check(self._lib.bds_acquire(self._h, C.byref(cs), p, n, int(is_complex), max_prn,
                                          carr.ctypes.data_as(_DP), cph.ctypes.data_as(_DP),
                                          pm.ctypes.data_as(_DP), det.ctypes.data_as(_IP)))
        return carr, cph, pm, det

    def frame_sync(self, signal, prns, prompt, cap=64):
        """bds_frame_sync: prompt [n_ch, n] -> (xcorr int32 [n_ch, M], [1-based index arrays])."""
        pr = np.ascontiguousarray(prompt, dtype=np.float64)
        n_ch, n = pr.shape
        sig = SIGNAL[str(signal).upper()]
        m = 1800 if sig == SIGNAL["B1C"] else 120
        M = max(n, m)
        prn = np.ascontiguousarray(prns, dtype=np.int32)
        xc = np.zeros((n_ch, M), dtype=np.int32)
        idx = np.zeros((n_ch, cap), dtype=np.int32)
        cnt = np.zeros(n_ch, dtype=np.int32)
        self._check(self._lib.bds_frame_sync(self._h, sig, n_ch, prn.ctypes.data_as(_IP), pr.ctypes.data_as(_DP), n,
                                             xc.ctypes.data_as(_IP), idx.ctypes.data_as(_IP), cnt.ctypes.data_as(_IP), cap))
        if np.any(cnt > cap):
            return self.frame_sync(signal, prns, prompt, cap=int(cnt.max()))
        return xc, [idx[c, :cnt[c]].copy() for c in range(n_ch)]

    def unpack_cplx(self, data):
        """bds_unpack_cplx: uint8[n] packed samples -> int8[4n] I/Q pairs (unpack_cplx.m)."""
        d = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        out = np.empty(4 * d.size, dtype=np.int8)
        self._check(self._lib.bds_unpack_cplx(self._h, d.ctypes.data_as(C.POINTER(C.c_uint8)), d.size,
                                              out.ctypes.data_as(C.POINTER(C.c_int8))))
        return out

    def unpack_cplx_file(self, filename_in, filename_out):
        self._check(self._lib.bds_unpack_cplx_file(self._h, os.fsencode(filename_in), os.fsencode(filename_out)))

    def acq_grid(self, n_prn, n_bins):
        rm = np.zeros(n_prn * n_bins, dtype=np.float32)
        ra = np.zeros(n_prn * n_bins, dtype=np.int32)
        self._check(self._lib.bds_acq_grid(self._h, rm.ctypes.data_as(C.POINTER(C.c_float)),
                                           ra.ctypes.data_as(_IP), rm.size))
        return rm.reshape(n_prn, n_bins), ra.reshape(n_prn, n_bins)

    def acq_candidates(self, prn):
        """(bin, codePhase) cells of `prn` the last run refined in f64, 1-based: int array [n, 2]."""
        n = self._check(self._lib.bds_acq_candidates(self._h, int(prn), None, None, 0))
        b = np.zeros(max(n, 1), dtype=np.int32)
        l = np.zeros(max(n, 1), dtype=np.int64)
        self._check(self._lib.bds_acq_candidates(self._h, int(prn), b.ctypes.data_as(_IP),
                                                 l.ctypes.data_as(C.POINTER(C.c_int64)), n))
        return np.stack([b[:n].astype(np.int64), l[:n]], axis=1)

    def acq_peaks(self, max_prn):
        pk = np.zeros(max_prn)
        dn = np.zeros(max_prn)
        fb = np.zeros(max_prn, dtype=np.int32)
        self._check(self._lib.bds_acq_peaks(self._h, max_prn, pk.ctypes.data_as(_DP), dn.ctypes.data_as(_DP),
                                            fb.ctypes.data_as(_IP)))
        return pk, dn, fb

    def acq_coherent_sums(self, settings, prn, phase, freqs, mode):
        """f64 coherent sums of caller-chosen cells (bds_acq_coherent_sums): complex array, mode 0 [nf, ncomp],
        modes 1 / 2 [segments * components, nf]."""
        cs = pack_settings(settings)
        fr = np.ascontiguousarray(freqs, dtype=np.float64)
        cap = max(int(getattr(settings, "fineNoncoh", 1) or 1), 1) * 2 * max(len(fr), 1)  # segments x components x frequencies
        out = np.zeros(2 * cap)
        n = self._check(self._lib.bds_acq_coherent_sums(self._h, C.byref(cs), int(prn), int(phase), fr.ctypes.data_as(_DP), len(fr),
                                                        int(mode), out.ctypes.data_as(_DP), cap))
        z = out[:2 * n:2] + 1j * out[1:2 * n:2]
        return z.reshape(len(fr), -1) if mode == 0 else z.reshape(-1, len(fr))  # modes 1 / 2: rows = (segment, component)

    def timing(self) -> dict:
        t = Timing()
        self._check(self._lib.bds_get_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in Timing._fields_ if not f.startswith("reserved")}

    # -- tracking --------------------------------------------------------------------
    def track(self, settings, source, channels, n_epochs, n_cno, fields):
        """source: file path (str/bytes) or int8 array of raw file bytes.
        Returns dict field -> array [n_ch, n_epochs] (C/N0 fields [n_ch, n_cno])."""
        cs = pack_settings(settings)
        nch = len(channels)
        carr = (Channel * nch)()
        for i, ch in enumerate(channels):
            carr[i].PRN = int(ch.PRN)
            carr[i].status = ord(ch.status) if isinstance(ch.status, str) else int(ch.status)
            carr[i].acquiredFreq = float(ch.acquiredFreq)
            carr[i].codePhase = float(ch.codePhase)
            carr[i].codeFreq = float(ch.codeFreq)
        out = TrackOut()
        out.n_ch, out.n_epochs, out.n_cno = nch, n_epochs, n_cno
        arrays = {}
        for f in fields:
            n = n_cno if f in ("DataCNo", "DataPLD", "PilotCNo", "PilotPLD", "SigCNo") else n_epochs
            arrays[f] = np.zeros((nch, n))
            setattr(out, f, arrays[f].ctypes.data_as(_DP))
        completed = np.zeros(nch, dtype=np.int32)
        status = np.zeros(nch, dtype=np.int32)
        out.completed = completed.ctypes.data_as(_IP)
        out.status = status.ctypes.data_as(_IP)
        if isinstance(source, (str, bytes, os.PathLike)):
            path = os.fsencode(source)
            self._check(self._lib.bds_track(self._h, C.byref(cs), path, nch, carr, C.byref(out)))
        else:
            a, p = _i8(source)
            self._check(self._lib.bds_track_mem(self._h, C.byref(cs), p, a.size, nch, carr, C.byref(out)))
        arrays["completed"] = completed
        arrays["status"] = status
        return arrays

    def _track_out(self, nch, n_epochs, n_cno, fields):
        out = TrackOut()
        out.n_ch, out.n_epochs, out.n_cno = nch, n_epochs, n_cno
        arrays = {}
        for f in fields:
            n = n_cno if f in ("DataCNo", "DataPLD", "PilotCNo", "PilotPLD", "SigCNo") else n_epochs
            arrays[f] = np.zeros((nch, n))
            setattr(out, f, arrays[f].ctypes.data_as(_DP))
        arrays["completed"] = np.zeros(nch, dtype=np.int32)
        arrays["status"] = np.zeros(nch, dtype=np.int32)
        out.completed = arrays["completed"].ctypes.data_as(_IP)
        out.status = arrays["status"].ctypes.data_as(_IP)
        return out, arrays

    def pre_run_device(self, settings, carr_freq, code_phase, peak_metric):
        """bds_pre_run_device: preRun.m:61-76 as a device kernel; returns the ctypes channel array."""
        cs = pack_settings(settings)
        a = np.ascontiguousarray(carr_freq, dtype=np.float64)
        b = np.ascontiguousarray(code_phase, dtype=np.float64)
        c = np.ascontiguousarray(peak_metric, dtype=np.float64)
        ch = (Channel * int(settings.numberOfChannels))()
        self._check(self._lib.bds_pre_run_device(self._h, C.byref(cs), a.size, a.ctypes.data_as(_DP), b.ctypes.data_as(_DP),
                                                 c.ctypes.data_as(_DP), ch))
        return ch

    def acquire_track(self, settings, samples, is_complex, path, n_epochs, n_cno, fields):
        """bds_acquire_track: acquisition -> device preRun -> tracking of the record at `path` in one native call.
        Returns ((carrFreq, codePhase, peakMetric, detected), channel array, dict of trackResults arrays)."""
        cs = pack_settings(settings)
        a, p = _i8(samples)
        n = a.size // 2 if is_complex else a.size
        max_prn = max(int(q) for q in np.atleast_1d(settings.acqSatelliteList))
        carr, cph, pm = np.zeros(max_prn), np.zeros(max_prn), np.zeros(max_prn)
        det = np.zeros(max_prn, dtype=np.int32)
        nch = int(settings.numberOfChannels)
        ch = (Channel * nch)()
        out, arrays = self._track_out(nch, n_epochs, n_cno, fields)
        self._check(self._lib.bds_acquire_track(self._h, C.byref(cs), p, n, int(bool(is_complex)), max_prn, carr.ctypes.data_as(_DP),
                                                cph.ctypes.data_as(_DP), pm.ctypes.data_as(_DP), det.ctypes.data_as(_IP),
                                                os.fsencode(path), ch, C.byref(out)))
        return (carr, cph, pm, det), ch, arrays

    def track_loaded_bytes(self) -> int:
        """Bytes of the record the last track() call copied to HBM (the window the channels can touch)."""
        return int(self._lib.bds_track_loaded_bytes(self._h))

    def track_correlate(self, settings, file_bytes, prns, state6):
        cs = pack_settings(settings)
        a, p = _i8(file_bytes)
        prn = np.ascontiguousarray(prns, dtype=np.int32)
        st = np.ascontiguousarray(state6, dtype=np.float64).reshape(prn.size, 6)
        sums = np.zeros((prn.size, 18))
        self._check(self._lib.bds_track_correlate(self._h, C.byref(cs), p, a.size, prn.size,
                                                  prn.ctypes.data_as(_IP), st.ctypes.data_as(_DP),
                                                  sums.ctypes.data_as(_DP)))
        return sums


def calc_loop_coef(lbw, zeta, k):
    t1, t2 = C.c_double(), C.c_double()
    lib().bds_calc_loop_coef(lbw, zeta, k, C.byref(t1), C.byref(t2))
    return t1.value, t2.value


def calc_loop_coef_carr(settings):
    cs = pack_settings(settings)
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().bds_calc_loop_coef_carr(C.byref(cs), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def calc_weighing_factor(settings):
    cs = pack_settings(settings)
    return float(lib().bds_calc_weighing_factor(C.byref(cs)))


def resample_plan(settings):
    """(new_fs, new_if, (wp1, wp2)) of the acquisition's resampling branch, or None when it is not taken
    (acquisition.m:54-55,66,103,119)."""
    cs = pack_settings(settings)
    fs, fi, wp = C.c_double(), C.c_double(), (C.c_double * 2)()
    rc = lib().bds_resample_plan(C.byref(cs), C.byref(fs), C.byref(fi), wp)
    if rc < 0:
        raise BdsError(rc, "bds_resample_plan")
    return (fs.value, fi.value, (wp[0], wp[1])) if rc else None


def fir1_bandpass(n_taps, wp1, wp2):
    """b = fir1(n_taps - 1, [wp1 wp2])."""
    b = np.zeros(n_taps)
    rc = lib().bds_fir1_bandpass(int(n_taps), float(wp1), float(wp2), b.ctypes.data_as(_DP))
    if rc < 0:
        raise BdsError(rc, "bds_fir1_bandpass")
    return b


def pre_run(settings, carr_freq, code_phase, peak_metric):
    cs = pack_settings(settings)
    n = len(carr_freq)
    nch = int(settings.numberOfChannels)
    ch = (Channel * nch)()
    a = np.ascontiguousarray(carr_freq, dtype=np.float64)
    b = np.ascontiguousarray(code_phase, dtype=np.float64)
    c = np.ascontiguousarray(peak_metric, dtype=np.float64)
    rc = lib().bds_pre_run(C.byref(cs), n, a.ctypes.data_as(_DP), b.ctypes.data_as(_DP), c.ctypes.data_as(_DP), ch)
    if rc < 0:
        raise BdsError(rc, "bds_pre_run")
    return ch
