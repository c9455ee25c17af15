"""GPU: the f64 coherent sums the acquisition decides on (csrc/bds_acq_corr.h, k_corr<NC, FM, KIND>) through the check entry
bds_acq_coherent_sums, against a direct float64 evaluation of the reference's expressions.

  * coarse cells (B2a/acquisition.m:194-209, B1C/acquisition.m:198-219): circular, both components in one pass;
  * fine-search blocks (B2a :287-316, B1C :253-287): the multi-frequency pass (six frequencies, both components per pass) and the
    one-frequency pass must agree to rounding (the two forms only shared result digests), and both with NumPy;
  * real and interleaved I/Q records (the record type is a template parameter of the kernel).
Tolerance: 1e-10 of max(largest sum, sqrt(sum |x|^2)) -- the size of a random-sign sum of the block.  The floor of any such
comparison is the carrier argument itself: f t / fs reaches ~1e5 cycles, whose f64 ulp is 1.5e-11 cycles = 1e-10 rad, and the
reference's own exp(1i f phasePoints) carries that noise sample by sample, while a phasor advanced by rotation carries the noise
of its last exact value (every 16th / 64th step).  Measured: <= 1.3e-11 against NumPy, ~2e-12 between the two passes."""
import numpy as np
import pytest

import bds_amd
from oracle import acquisition as oacq
from oracle import codes

from helpers import as_complex, cfg1_b2a, cfg1_b2a_iq, small_b1c, small_b1c_iq, spc_of

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _b2a_fine_ref(x, s, prn, phase, freqs):
    spc = spc_of(s)
    ts = 1.0 / s.samplingFreq
    nn = int(s.fineNoncoh) * spc
    cvi = np.floor((ts * np.arange(1, nn + 1, dtype=np.float64)) / (1.0 / s.codeFreqBasis)).astype(np.int64)
    cidx = np.fmod(cvi, int(s.codeLength))
    long_c = [codes.generate_b2a_data_code(prn, s)[cidx], codes.generate_b2a_pilot_code(prn, s)[cidx]]
    sig = x[phase - 1: phase - 1 + nn]
    t = np.tile(np.arange(spc, dtype=np.float64), int(s.fineNoncoh))  # the library restarts the phase at every segment (|.| per segment)
    out = np.zeros((int(s.fineNoncoh), 2, len(freqs)), dtype=np.complex128)
    for k, f in enumerate(freqs):
        carr = np.exp(2j * np.pi * ((f * (t * ts)) % 1.0))
        for c in range(2):
            out[:, c, k] = (long_c[c] * carr * sig).reshape(int(s.fineNoncoh), spc).sum(axis=1)
    return out.reshape(-1, len(freqs))


def _scale(ref, block):
    return max(np.abs(ref).max(), float(np.sqrt(np.sum(np.abs(block) ** 2))))


def _check_fine(ctx, s, x, prn, phase, freqs, ref, block):
    multi = ctx.acq_coherent_sums(s, prn, phase, freqs, 1)
    single = ctx.acq_coherent_sums(s, prn, phase, freqs, 2)
    scale = _scale(ref, block)
    assert multi.shape == ref.shape == single.shape
    assert np.abs(multi - ref).max() <= TOL * scale, np.abs(multi - ref).max() / scale
    assert np.abs(single - ref).max() <= TOL * scale, np.abs(single - ref).max() / scale
    assert np.abs(multi - single).max() <= TOL * scale, np.abs(multi - single).max() / scale
    # what the fine search ranks: the non-coherent sum over segments and components, per frequency
    assert np.argmax(np.abs(multi).sum(axis=0)) == np.argmax(np.abs(ref).sum(axis=0)) == np.argmax(np.abs(single).sum(axis=0))


@pytest.mark.parametrize("iq", [False, True])
def test_b2a_fine_block_multi_single_numpy(ctx, iq):
    s, x, _ = cfg1_b2a_iq() if iq else cfg1_b2a()
    xs = as_complex(x) if iq else x.astype(np.float64)
    got = bds_amd.acquisition(xs if iq else x, s, verbose=False)
    phase = int(got.codePhase[18])
    assert phase > 0
    fb = oacq.freq_bins(s)
    freqs = fb[len(fb) // 2] - s.acqStep / 2 + 25.0 * np.arange(17)  # :300-301 (17 = three passes of six)
    spc = spc_of(s)
    _check_fine(ctx, s, xs, 19, phase, freqs, _b2a_fine_ref(xs, s, 19, phase, freqs), xs[phase - 1: phase - 1 + spc])
    # a ragged frequency count (one pass of four) and another start
    _check_fine(ctx, s, xs, 19, 1, freqs[:4], _b2a_fine_ref(xs, s, 19, 1, freqs[:4]), xs[:spc])


@pytest.mark.parametrize("iq", [False, True])
def test_b2a_coarse_cells(ctx, iq):
    s, x, _ = cfg1_b2a_iq() if iq else cfg1_b2a()
    xs = as_complex(x) if iq else x.astype(np.float64)
    bds_amd.acquisition(xs if iq else x, s, verbose=False)
    spc = spc_of(s)
    n = 2 * spc
    tabs = [codes.make_b2a_data_table(19, s), codes.make_b2a_pilot_table(19, s)]
    freqs = oacq.freq_bins(s)
    ts = 1.0 / s.samplingFreq
    for phase in (1, 36771, n - 5, n):  # the last two wrap within a few samples
        z = ctx.acq_coherent_sums(s, 19, phase, freqs, 0)
        idx = (phase - 1 + np.arange(spc)) % n
        ref = np.zeros((len(freqs), 2), dtype=np.complex128)
        for k, f in enumerate(freqs):
            carr = np.exp(2j * np.pi * ((f * (idx.astype(np.float64) * ts)) % 1.0))
            for c in range(2):
                ref[k, c] = np.sum(xs[idx] * tabs[c] * carr)
        scale = _scale(ref, xs[idx])
        assert z.shape == ref.shape
        assert np.abs(z - ref).max() <= TOL * scale, (phase, np.abs(z - ref).max() / scale)


@pytest.mark.parametrize("iq", [False, True])
def test_b1c_fine_block_and_coarse_cells(ctx, iq):
    s, x, _ = small_b1c_iq() if iq else small_b1c()
    xs = as_complex(x) if iq else x.astype(np.float64)
    got = bds_amd.acquisition(xs if iq else x, s, verbose=False)
    phase = int(got.codePhase[2])
    assert phase > 0
    spc = spc_of(s)
    ts = 1.0 / s.samplingFreq
    ncomp = 2 if s.pilotACQflag == 1 else 1
    tabs = [codes.make_data_table(s, 3)] + ([codes.make_pilot_table(s, 3)] if ncomp == 2 else [])
    fb = oacq.freq_bins(s)
    nfine = int(round(s.acqStep / 25)) * 2 + 1
    freqs = fb[len(fb) // 2] - s.acqStep + 25.0 * np.arange(nfine)  # :282-283
    sig0 = xs[phase - 1: phase - 1 + spc]
    sig0 = sig0 - np.mean(sig0)  # :253-254
    t = np.arange(spc, dtype=np.float64)
    ref = np.zeros((ncomp, len(freqs)), dtype=np.complex128)
    for k, f in enumerate(freqs):
        carr = np.exp(2j * np.pi * ((f * (t * ts)) % 1.0))
        for c in range(ncomp):
            ref[c, k] = np.sum(sig0 * tabs[c] * carr)
    _check_fine(ctx, s, xs, 3, phase, freqs, ref, sig0)
    # coarse cells: X = the coherent block, N = X + one code period (B1C/acquisition.m:131-141)
    _, x_len, n = oacq._b1c_sizes(s)
    for ph in (phase, n - 3):
        z = ctx.acq_coherent_sums(s, 3, ph, fb[:3], 0)
        idx = (ph - 1 + np.arange(x_len)) % n
        refc = np.zeros((3, ncomp), dtype=np.complex128)
        for k, f in enumerate(fb[:3]):
            carr = np.exp(2j * np.pi * ((f * (idx.astype(np.float64) * ts)) % 1.0))
            for c in range(ncomp):
                tab = np.resize(tabs[c], x_len) if len(tabs[c]) < x_len else tabs[c][:x_len]
                refc[k, c] = np.sum(xs[idx] * tab * carr)
        scale = _scale(refc, xs[idx])
        assert np.abs(z - refc).max() <= TOL * scale, (ph, np.abs(z - refc).max() / scale)


def test_b2a_low_rate_short_slices(ctx):
    """fs = 10.77 MS/s: a segment is 10 770 samples, the eight slices of a job are 1 536 samples and the last one 18 -- fewer
    live lanes in its only wave than the six that evaluate a block's exact phasors (the first quad-load version of k_corr left
    the last frequency of a pass unset there: carrFreq 75 Hz off on one PRN of test_every_specialised_plan_pair[768])."""
    from bds_amd import synth
    s = bds_amd.init_settings_b2a(samplingFreq=10.0e6 + 1000.0 * 770, IF=2.5e6, acqSatelliteList=[7, 19, 33], acqSearchBand=600,
                                  acqStep=200, fineNoncoh=4)
    spc = spc_of(s)
    sats = [synth.Sat(19, 130.0, 0.41 * spc, 0.7, 47.0), synth.Sat(33, -90.0, 0.83 * spc, 2.2, 45.0)]
    x = synth.make_if(s, sats, 7 * spc, seed=1270)
    xs = x.astype(np.float64)
    got = bds_amd.acquisition(x, s, verbose=False)
    fb = oacq.freq_bins(s)
    for prn in (19, 33):
        phase = int(got.codePhase[prn - 1])
        assert phase > 0
        freqs = fb[len(fb) // 2] - s.acqStep / 2 + 25.0 * np.arange(9)  # a pass of six and a pass of three
        _check_fine(ctx, s, xs, prn, phase, freqs, _b2a_fine_ref(xs, s, prn, phase, freqs), xs[phase - 1: phase - 1 + spc])
