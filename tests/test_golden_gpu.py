"""GPU parity against the committed golden vectors (inputs and expected outputs in tests/golden)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

import bds_amd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return z, bds_amd.Settings(**json.loads(str(z["settings"])))


@pytest.mark.parametrize("name", ["acq_b2a_small", "acq_b1c_small"])
def test_acquisition_vs_golden(ctx, name):
    z, s = load(name)
    r = bds_amd.acquisition(z["x"], s)
    np.testing.assert_array_equal(r.codePhase, z["codePhase"])
    np.testing.assert_array_equal(r.carrFreq, z["carrFreq"])
    np.testing.assert_allclose(r.peakMetric, z["peakMetric"], rtol=1e-6)
    rm, ra = ctx.acq_grid(*z["row_max"].shape)
    np.testing.assert_allclose(rm, z["row_max"], rtol={0: 1e-5, 1: 1e-3}[ctx.timing()["half_storage"]])


@pytest.mark.parametrize("name", ["trk_b2a_small", "trk_nb_small", "trk_wb_small"])
def test_tracking_vs_golden(ctx, name):
    z, s = load(name)
    chans = [SimpleNamespace(**c) for c in json.loads(str(z["channels"]))]
    mode = str(z["mode"])
    got, _ = bds_amd.tracking(z["x"], chans, s, mode=mode)
    p = np.hypot(z["I_P"], z["Q_P"]).max()
    for f in ("I_P", "Q_P", "I_E", "Q_L", "Pilot_I_P", "Pilot_Q_P"):
        np.testing.assert_allclose(np.stack([getattr(g, f) for g in got]), z[f], rtol=0, atol=1e-4 * p, err_msg=f)
    np.testing.assert_array_equal(np.stack([g.absoluteSample for g in got]), z["absoluteSample"])
    np.testing.assert_allclose(np.stack([g.carrFreq for g in got]), z["carrFreq"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(np.stack([g.codeFreq for g in got]), z["codeFreq"], rtol=0, atol=1e-6)
    # open loop against the stored raw correlator sums of the first epoch
    st = [[z["absoluteSample"][c, 0], z["blk"][c, 0], z["remCodePhase"][c, 0], z["codeFreq"][c, 0],
           z["remCarrPhase"][c, 0], z["carrFreq"][c, 0]] for c in range(len(chans))]
    sums = ctx.track_correlate(s, z["x"], [c.PRN for c in chans], st)
    for c in range(len(chans)):
        pc = np.hypot(z["raw_sums"][c, 0, 2], z["raw_sums"][c, 0, 3])
        np.testing.assert_allclose(sums[c], z["raw_sums"][c, 0], rtol=0, atol=1e-6 * pc)
