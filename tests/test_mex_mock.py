"""mex/bds_mex.c EXECUTED: the gateway compiled for real against a small stand-in for the MEX runtime
(tests/mex_stub/mex_mock.c: arrays, structs, strings, error exit by longjmp) and driven through ctypes with what MATLAB's
wrappers (mex/bds_acquire_common.m, bds_track_common.m) would hand to mexFunction.  Not MATLAB -- there is none in the image --
but every line of the gateway runs: settings packing and its error messages, argument checks, output shapes (column-major
[epochs x channels]), and the results are compared with the ctypes host path on the same inputs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import bds_amd
from bds_amd import native

from helpers import as_complex, cfg1_b2a, track_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "mex_stub")
SO = os.path.join(STUB, "_build", "libbds_mex_mock.so")
P = ctypes.c_void_p


@pytest.fixture(scope="module")
def mex():
    pkg = os.path.dirname(native._LIB_PATH)
    libname = os.path.splitext(os.path.basename(native._LIB_PATH))[0][3:]  # the library the suite runs on (hooks build)
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-shared", "-fPIC", "-I", STUB,
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "mex", "bds_mex.c"), os.path.join(STUB, "mex_mock.c"),
                           "-L", pkg, "-l" + libname, "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", SO])
    L = ctypes.CDLL(SO)
    for name, res, args in [
        ("mxCreateDoubleMatrix", P, [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]),
        ("mxCreateNumericMatrix", P, [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]),
        ("mxCreateStructMatrix", P, [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, P]),
        ("mxCreateString", P, [ctypes.c_char_p]), ("mxCreateLogicalScalar", P, [ctypes.c_bool]),
        ("mxAddField", ctypes.c_int, [P, ctypes.c_char_p]), ("mxSetField", None, [P, ctypes.c_size_t, ctypes.c_char_p, P]),
        ("mxGetField", P, [P, ctypes.c_size_t, ctypes.c_char_p]), ("mxGetDoubles", ctypes.POINTER(ctypes.c_double), [P]),
        ("mxGetInt8s", ctypes.POINTER(ctypes.c_int8), [P]), ("mxGetInt32s", ctypes.POINTER(ctypes.c_int32), [P]),
        ("mxGetNumberOfElements", ctypes.c_size_t, [P]), ("mxDestroyArray", None, [P]),
        ("mock_call", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(P), ctypes.c_int, ctypes.POINTER(P)]),
        ("mock_error_id", ctypes.c_char_p, []), ("mock_error_msg", ctypes.c_char_p, []), ("mock_run_atexit", None, []),
        ("mock_rows", ctypes.c_size_t, [P]), ("mock_cols", ctypes.c_size_t, [P]), ("mock_nfields", ctypes.c_int, [P]),
        ("mock_field_name", ctypes.c_char_p, [P, ctypes.c_int]),
    ]:
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    yield Mex(L)
    L.mock_run_atexit()  # what clearing the MEX file does: bds_multi_destroy


class MexError(Exception):
    def __init__(self, ident, msg):
        super().__init__(f"{ident}: {msg}")
        self.ident, self.msg = ident, msg


class Mex:
    def __init__(self, L):
        self.L = L

    # ---- MATLAB values -> mxArray ----
    def double(self, v):
        v = np.atleast_1d(np.asarray(v, dtype=np.float64))
        a = self.L.mxCreateDoubleMatrix(1, v.size, 0)
        if v.size:
            ctypes.memmove(self.L.mxGetDoubles(a), v.ctypes.data, 8 * v.size)
        return a

    def int8(self, v):
        v = np.ascontiguousarray(v, dtype=np.int8)
        a = self.L.mxCreateNumericMatrix(1, v.size, 8, 0)
        ctypes.memmove(self.L.mxGetInt8s(a), v.ctypes.data, v.size)
        return a

    def value(self, v):
        if isinstance(v, str):
            return self.L.mxCreateString(v.encode())
        if isinstance(v, bool):
            return self.L.mxCreateLogicalScalar(v)
        return self.double(v)

    def struct(self, elems):
        """1 x n struct array from a list of dicts with the same keys."""
        names = list(elems[0])
        s = self.L.mxCreateStructMatrix(1, len(elems), 0, None)
        for n in names:
            self.L.mxAddField(s, n.encode())
        for i, e in enumerate(elems):
            for n in names:
                self.L.mxSetField(s, i, n.encode(), self.value(e[n]))
        return s

    def settings(self, s, drop=(), **extra):
        d = {k: v for k, v in vars(s).items() if k != "signal" and k not in drop}
        d.update(extra)
        return self.struct([d])

    # ---- the call ----
    def call(self, nlhs, *args):
        prhs = (P * len(args))(*args)
        plhs = (P * max(nlhs, 1))()
        rc = self.L.mock_call(nlhs, plhs, len(args), prhs)
        for a in args:
            self.L.mxDestroyArray(a)
        if rc:
            raise MexError(self.L.mock_error_id().decode(), self.L.mock_error_msg().decode())
        return [plhs[i] for i in range(max(nlhs, 1))]

    def doubles(self, a):
        m, n = self.L.mock_rows(a), self.L.mock_cols(a)
        out = np.ctypeslib.as_array(self.L.mxGetDoubles(a), shape=(m * n,)).copy() if m * n else np.zeros(0)
        return out.reshape((n, m)).T  # column-major -> [m, n]

    def int32s(self, a):
        n = self.L.mxGetNumberOfElements(a)
        return np.ctypeslib.as_array(self.L.mxGetInt32s(a), shape=(n,)).copy()

    def fields(self, a):
        return [self.L.mock_field_name(a, k).decode() for k in range(self.L.mock_nfields(a))]


SIG = {"B1C": 1, "B2A": 2}


# ---------------------------------------------------------------- no GPU needed ----------------------------------------
def test_gen_code_through_the_gateway(mex):
    for signal, kname, prn in (("B2A", "data", 19), ("B2A", "pilot", 63), ("B1C", "data_boc11", 1), ("B1C", "pilot_boc61", 30)):
        kind = native.CODE_KIND[kname]
        (out,) = mex.call(1, mex.value("gen_code"), mex.double(SIG[signal]), mex.double(kind), mex.double(prn))
        got = mex.doubles(out)[0]
        mex.L.mxDestroyArray(out)
        want = native.gen_code(signal, kname, prn)
        np.testing.assert_array_equal(got, np.asarray(want, dtype=np.float64))
        assert got.size == native.CODE_LEN[kind] and set(np.unique(got)) <= {-1.0, 1.0}


def test_settings_errors_name_the_field(mex):
    s = bds_amd.init_settings_b2a()
    x = np.zeros(16, dtype=np.int8)
    with pytest.raises(MexError, match=r"settings\.acqStep is missing") as e:
        mex.call(4, mex.value("acquire"), mex.int8(x), mex.settings(s, drop=("acqStep",)), mex.double(2))
    assert e.value.ident == "bds:settings"
    with pytest.raises(MexError, match=r"settings\.IF must be a numeric scalar"):
        mex.call(4, mex.value("acquire"), mex.int8(x), mex.settings(s, IF=[1.0, 2.0]), mex.double(2))
    with pytest.raises(MexError, match="dataType must be 'schar'"):
        mex.call(4, mex.value("acquire"), mex.int8(x), mex.settings(s, dataType="int16"), mex.double(2))
    with pytest.raises(MexError, match="signal must be 1"):
        mex.call(4, mex.value("acquire"), mex.int8(x), mex.settings(s), mex.double(3))
    with pytest.raises(MexError, match=r"settings\.FEBW is missing"):  # a B2a struct handed to the B1C receiver
        mex.call(4, mex.value("acquire"), mex.int8(x), mex.settings(s, acqCohT=10, pilotACQflag=1), mex.double(1))
    with pytest.raises(MexError, match="int8 longSignal"):  # double longSignal: the wrapper casts, the gateway insists
        mex.call(4, mex.value("acquire"), mex.double(x), mex.settings(s), mex.double(2))
    with pytest.raises(MexError, match="unknown command nonsense"):
        mex.call(1, mex.value("nonsense"))
    with pytest.raises(MexError, match="first argument"):
        mex.call(1, mex.double(1.0))
    with pytest.raises(MexError, match="bad signal/kind/prn"):
        mex.call(1, mex.value("gen_code"), mex.double(2), mex.double(0), mex.double(64))


# ---------------------------------------------------------------- on the GPU --------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("iq", [False, True])
def test_acquire_through_the_gateway(mex, iq):
    """[carrFreq, codePhase, peakMetric, detected] = bds_mex('acquire', int8(longSignal), settings, signal, iq) -- what
    mex/bds_acquire_common.m:16 calls -- against bds_amd.acquisition on the same block (BASELINE.json configs[0])."""
    from helpers import cfg1_b2a_iq

    if iq:
        s, x, _ = cfg1_b2a_iq()  # int8 I/Q pairs, interleaved: what bds_acquire_common.m rebuilds from the complex longSignal
        want = bds_amd.acquisition(as_complex(x), s, verbose=False)
        args = [mex.value("acquire"), mex.int8(x), mex.settings(s), mex.double(2), mex.value(True)]
    else:
        s, x, _ = cfg1_b2a()
        want = bds_amd.acquisition(x, s, verbose=False)
        args = [mex.value("acquire"), mex.int8(x), mex.settings(s), mex.double(2)]
    cf, cp, pm, det = mex.call(4, *args)
    got = [mex.doubles(a) for a in (cf, cp, pm)]
    d = mex.int32s(det)
    for a in (cf, cp, pm, det):
        mex.L.mxDestroyArray(a)
    n = max(s.acqSatelliteList)
    assert all(g.shape == (1, n) for g in got) and d.shape == (n,)
    np.testing.assert_array_equal(got[0][0], want.carrFreq)
    np.testing.assert_array_equal(got[1][0], want.codePhase)
    np.testing.assert_array_equal(got[2][0], want.peakMetric)
    np.testing.assert_array_equal(d != 0, want.carrFreq != 0)
    assert np.count_nonzero(want.carrFreq) >= 1


@pytest.mark.gpu
def test_acquire_reports_library_errors(mex):
    s = bds_amd.init_settings_b2a(acqSatelliteList=[5])
    with pytest.raises(MexError, match="acquisition needs at least") as e:
        mex.call(4, mex.value("acquire"), mex.int8(np.zeros(1000, dtype=np.int8)), mex.settings(s), mex.double(2))
    assert e.value.ident == "bds:acquire"


@pytest.mark.gpu
@pytest.mark.parametrize("signal,mode,n_epochs", [("B2A", "B2A", 40), ("B1C", "NB", 8), ("B1C", "WB", 8)])
def test_track_through_the_gateway(mex, tmp_path, signal, mode, n_epochs):
    """out = bds_mex('track', path, channel, settings, signal) -- mex/bds_track_common.m:5 -- : the field set of the variant, every
    array [epochs x channels] column-major as the wrapper slices it (out.(n)(:, ch).'), against bds_amd.tracking on the same file."""
    s, x, chans = track_case(signal, mode, n_epochs)
    path = str(tmp_path / "rec.bin")
    x.tofile(path)
    want, _ = bds_amd.tracking(path, chans, s, mode=mode)
    ch = mex.struct([dict(PRN=c.PRN, acquiredFreq=c.acquiredFreq, codePhase=c.codePhase, codeFreq=c.codeFreq, status=c.status) for c in chans])
    (out,) = mex.call(1, mex.value("track"), mex.value(path), ch, mex.settings(s), mex.double(SIG[signal]))
    names = mex.fields(out)
    from bds_amd.tracking import field_set

    n, m, ep, cno_g, _ = field_set(s, mode)  # (the gateway's names: SigCNo is what the wrapper renames to B1C_CNo / B2a_CNo)
    assert names == ep + cno_g + ["completed", "status"]
    for f in ep + cno_g:
        a = mex.doubles(mex.L.mxGetField(out, 0, f.encode()))
        assert a.shape == ((n if f in ep else m), len(chans))
        src = {"SigCNo": "B2a_CNo" if mode == "B2A" else "B1C_CNo"}.get(f, f)
        for c, w in enumerate(want):
            np.testing.assert_array_equal(a[:, c], getattr(w, src), err_msg=f)
    comp = mex.int32s(mex.L.mxGetField(out, 0, b"completed"))
    stat = mex.int32s(mex.L.mxGetField(out, 0, b"status"))
    assert list(comp) == [n] * len(chans) and [chr(v) for v in stat] == [w.status for w in want]
    mex.L.mxDestroyArray(out)


@pytest.mark.gpu
def test_frame_sync_through_the_gateway(mex):
    """[XcorrResult, index] = bds_mex('frame_sync', signal, PRN, bits): B2a preamble x NH pattern found where it was put."""
    rng = np.random.default_rng(4)
    pat = np.asarray(native.sync_pattern("B2A", 19), dtype=np.float64)
    bits = np.sign(rng.normal(size=1500))
    bits[300:300 + pat.size] = pat
    bits[900:900 + pat.size] = -pat
    xc, idx = mex.call(2, mex.value("frame_sync"), mex.double(2), mex.double(19), mex.double(bits * 3.7))
    x = mex.doubles(xc)[0]
    i = mex.doubles(idx)
    mex.L.mxDestroyArray(xc), mex.L.mxDestroyArray(idx)
    want_x, want_i = bds_amd.get_context(0).frame_sync("B2A", [19], (bits * 3.7)[None, :])
    np.testing.assert_array_equal(x, np.asarray(want_x[0], dtype=np.float64))
    assert i.shape[1] == 1 and list(i[:, 0]) == list(want_i[0]) and {301, 901} <= set(i[:, 0])
