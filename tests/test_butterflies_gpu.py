"""The FMA-folded butterflies of csrc/bds_fft_fma.h (the arithmetic core of both wave-private search passes) against a
double-precision DFT: tools/probe/bfly_check.hip is compiled with hipcc on the GPU box and run; it covers both directions,
16 and 8 points, with and without input twiddles, next to the plain Butterfly<16, DIR> they replace (1e-6 of the largest
output; fp32 rounding of a 16-point transform is ~1e-7)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fma_butterflies_against_double_dft(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = tmp_path / "bfly_check"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-unused-result",
           "-I" + os.path.join(ROOT, "bds-3-b1c-b2a-sdr-receiver_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "probe", "bfly_check.hip"), "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "max |err|" in l]
    assert len(lines) == 18 and r.stdout.strip().endswith("ok"), r.stdout
    for l in lines:
        assert float(l.split("=")[-1]) < 1e-6, l
