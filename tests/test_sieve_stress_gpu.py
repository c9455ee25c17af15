"""The sieve at the cfg3 plan (768 x 4096, all 63 PRNs x 201 bins) on inputs that stress the fp16-stored spectra: a CW
interferer at J/N = +40 dB (one spectral line carries the three fp16 roundings coherently: the worst case of tools/sieve_stress.py),
a 2-bit record (the unpack_cplx alphabet), a block clipped at +-127.  The search grid with fp16 storage is compared with the one
with fp32 storage: every row maximum at least 2x inside kDelta / 2 (= 2e-3, bds_acq.hip), the f64 peaks and acqResults of the
two runs identical bit for bit."""
import importlib.util
import os

import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("sieve_stress", os.path.join(ROOT, "tools", "sieve_stress.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_fp16_sieve_on_stress_inputs():
    st = _tool()
    s, _, sats, _ = bench.build_workload("b1c")
    prns = list(range(1, 64))
    blocks = st.blocks(s, sats, 4 * st.SPC)
    for name in ("cw+40", "2bit", "clipped"):
        x = blocks[name]
        h, ha, hp, hm, hres = st.run(s, x, prns, {})
        f, fa, fp, fm, fres = st.run(s, x, prns, {"BDS_ACQ_FP16": "0"})
        assert fm == 0
        if hm == 1:  # fp16 storage kept: its grid must sit well inside the tolerance it was searched with
            err = (np.abs(h - f) / f.max(axis=1, keepdims=True)).max()
            assert err < 1e-3, (name, err)
        assert np.array_equal(hp, fp), name
        for u, v in zip(hres, fres):
            assert np.array_equal(u, v), name
