"""The NumPy models of the two wave-private search passes (tools/proto_cols_wave.py, tools/proto_rows_wave.py) are the written
statement of the kernels' stage algebra, lane maps and LDS layouts: they must reproduce numpy.fft on random data and find
every LDS access class free of bank conflicts.  (CPU only; the kernels themselves are compared with the oracle in the GPU suite.)"""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("S", [256, 512, 768, 1024])
def test_column_pass_model(S):
    assert _load("proto_cols_wave").run(S, seed=S, verbose=False) < 1e-12


def test_row_pass_model():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "proto_rows_wave.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "'1a.write': 1, '1b.read': 1, '1b.write': 1, '2.read': 1" in r.stdout, r.stdout
