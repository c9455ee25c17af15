import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """Process-wide bds_ctx on GPU 0 (fails loudly when the HIP library or GPU is missing)."""
    import bds_amd

    return bds_amd.get_context(0)


def pytest_sessionfinish(session, exitstatus):
    """On the debug build of the library (BDS_DEBUG=1 ./build.sh, loaded through BDS_LIB_PATH: tools/run_debug.sh) every
    kernel checks its code-table / IF-window / candidate-list indices on the device: none may have failed."""
    try:
        from bds_amd import native
    except Exception:  # noqa: BLE001
        return
    n = native.debug_failures()
    if n:
        print(f"\nBDS_DEBUG: {n} device-side bounds checks FAILED during this session (see the BDS_DASSERT lines above)")
        session.exitstatus = 1
