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


# ---- the N-point plan of round 6 (tools/proto_pfa53.py; kernels: csrc/bds_acq_pfa.h) -------------------------------------------------
@pytest.mark.parametrize("dims", [(5, 4, 9), (53, 12, 125)])
def test_prime_factor_maps_and_the_rotation_identity(dims):
    """Good-Thomas: CRT order on the spectrum side, Ruritanian order on the lag side, no twiddles between coprime dimensions; a shift
    of the spectrum is a rotation by the same amount in every dimension (asserted inside check_maps)"""
    assert _load("proto_pfa53").check_maps(dims) < 1e-12


def test_bin_shift_identity_on_the_golden_block():
    """rows of the reference's statement (per-bin carrier, per-bin fft: oracle.acquisition.b1c_coarse_rows, B1C/acquisition.m:191-222)
    against ONE forward transform shifted per bin and the multi-dimensional inverse, on the committed golden block"""
    assert _load("proto_pfa53").check_golden_block() < 1e-9


def test_mfma_column_stage_model():
    """the 53 x 12 column stage as k_pfa_cols issues it -- fp16 data as the A operand, hi + lo fp16 coefficients, the fragment layouts
    of v_mfma_f32_16x16x32_f16 lane by lane, real 12-point transforms per lane, (re, im) lane pairs -- against a float64 transform;
    with the hi coefficients alone the 2^-12 coefficient rounding shows"""
    both, hi_only = _load("proto_pfa53").check_cols()
    assert both < 1e-6 and 1e-5 < hi_only < 1e-3


def test_hi_only_bound_pass_margin():
    """k_pfa_cols' first pass uses fp16(coefficient) alone and widens its Cauchy-Schwarz bound by 5e-4 sqrt(sum of the lag's outputs):
    the values' pass (hi + lo) never exceeds that, on random inputs and on inputs built against one output's lo column"""
    assert _load("proto_pfa53").check_bound() < 1.0


def test_3125_point_rows_model():
    assert _load("proto_pfa53").check_rows() < 1e-12
