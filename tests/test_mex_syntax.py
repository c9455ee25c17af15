"""mex/bds_mex.c has never met MATLAB's compiler (none exists in this image).  This compiles it with
-fsyntax-only against tests/mex_stub/mex.h (declarations of the MEX API functions it uses) and the real
include/bds_mi355x.h (tests/test_mex_mock.py goes further and executes the gateway against a mock MEX runtime) -- every library entry the gateway calls is type-checked against the
C ABI, and a field the gateway forgets to require would show up in the field-list check below."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gateway_compiles_against_the_c_abi():
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "tests", "mex_stub"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "mex", "bds_mex.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_gateway_requires_every_appendix_d_field():
    """every settings field the path reads is required (error names the field); none is
    silently defaulted.  dataType must be 'schar'."""
    src = open(os.path.join(ROOT, "mex", "bds_mex.c")).read()
    body = src[src.index("static void pack_settings"):src.index("static void do_acquire")]
    required = set(re.findall(r'(?:num|need)\(s, "(\w+)"\)', body))
    common = {"samplingFreq", "IF", "codeFreqBasis", "carrFreqBasis", "codeLength", "acqSatelliteList", "acqSearchBand", "acqStep",
              "acqThreshold", "resamplingThreshold", "resamplingflag", "fileType", "dataType", "skipNumberOfBytes", "msToProcess",
              "numberOfChannels", "intTime", "dllCorrelatorSpacing", "dllDampingRatio", "dllNoiseBandwidth", "pllNoiseBandwidth",
              "CNoInterval", "pilotTRKflag"}
    assert common | {"fineNoncoh", "acqCohT", "pilotACQflag", "FEBW"} == required
    assert "schar" in body and "field(" not in body  # no optional-with-default accessor left
