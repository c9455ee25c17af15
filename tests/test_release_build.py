"""The RELEASE library (libbds_mi355x.so: what bench.py, smoke() and a MATLAB host load) against the test-hooks build the
rest of the suite runs on: it reads five documented environment knobs and nothing else, so no stray BDS_* variable of a host
session can switch off the completeness self-check, widen the sieve tolerance, force a fallback or redirect dlopen."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bds-3-b1c-b2a-sdr-receiver_amd")
RELEASE = os.path.join(PKG, "libbds_mi355x.so")
HOOKS = os.path.join(PKG, "libbds_mi355x_hooks.so")
DOCUMENTED = {"BDS_ACQ_FP16", "BDS_TRK_PREC", "BDS_VERBOSE", "BDS_ACQ_CLOCKPROBE", "BDS_ACQ_PAIR_GB"}


def env_names(path):
    """BDS_* names among the printable strings of a shared library (what `strings lib | grep BDS_` shows)"""
    blob = open(path, "rb").read()
    return {m.decode() for m in re.findall(rb"BDS_[A-Z0-9_]{2,}", blob)}


def test_release_library_reads_only_the_documented_knobs():
    names = env_names(RELEASE)
    assert DOCUMENTED <= names
    # error-code / macro names may appear in messages; environment switches of the hooks build may not
    hooks_only = env_names(HOOKS) - names
    for n in ("BDS_ACQ_KDELTA", "BDS_ACQ_NO_SELFCHECK", "BDS_ACQ_TEST_FORCE_FALLBACK", "BDS_MULTI_TEST_ALIAS", "BDS_RCCL_LIB",
              "BDS_ACQ_FORCE_L1L2", "BDS_ACQ_WCOLS", "BDS_ACQ_WROWS", "BDS_TRK_PERSAMPLE", "BDS_TRK_NBLOCKS"):
        assert n in hooks_only, n
    assert len(names) <= 6, sorted(names)
    hdr = open(os.path.join(ROOT, "include", "bds_mi355x.h")).read()
    for n in DOCUMENTED:
        assert n in hdr, n


def test_build_flags():
    import ctypes

    assert ctypes.CDLL(RELEASE).bds_build_flags() == 0
    assert ctypes.CDLL(HOOKS).bds_build_flags() & 1


CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import bds_amd
from bds_amd import native
from helpers import medium_b2a
s, x, _ = medium_b2a()
r = bds_amd.acquisition(x, s)
tm = bds_amd.get_context(0).timing()
print(json.dumps({"hooks": native.has_test_hooks(), "half": int(tm["half_storage"]), "l1": int(tm["plan_l1"]), "cols": int(tm["cols_kernel"]),
                  "sha": hashlib.sha256(np.stack([r.carrFreq, r.codePhase, r.peakMetric]).tobytes()).hexdigest()}))
"""


def _child(lib, extra):
    env = dict(os.environ, BDS_LIB_PATH=lib, **extra)
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_release_library_ignores_the_test_switches():
    knobs = {"BDS_ACQ_TEST_FORCE_FALLBACK": "1", "BDS_ACQ_KDELTA": "0.5", "BDS_ACQ_WCOLS": "0", "BDS_ACQ_SMALL": "0", "BDS_ACQ_NO_SELFCHECK": "1"}
    plain = _child(RELEASE, {})
    knobbed = _child(RELEASE, knobs)
    assert not plain["hooks"] and plain == knobbed          # same plan, same kernels, same storage mode, same bits
    hooked = _child(HOOKS, knobs)
    assert hooked["hooks"] and (hooked["half"], hooked["l1"], hooked["cols"]) != (plain["half"], plain["l1"], plain["cols"])
    assert hooked["sha"] == plain["sha"]                     # the decisions do not depend on the path taken
