import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The suite runs on the TEST-HOOKS build of the library (libbds_mi355x_hooks.so, built beside the release library by
# build.sh): the kernel-selection / launch-shape / sieve switches the tests flip do not exist in the release library.
# tests/test_release_build.py and bench.py (also when started from tests/test_bench_gpu.py) load the release library.
_PKG = os.path.join(ROOT, "bds-3-b1c-b2a-sdr-receiver_amd")
os.environ.setdefault("BDS_LIB_PATH", os.path.join(_PKG, "libbds_mi355x_hooks.so"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """Process-wide bds_ctx on GPU 0 (fails loudly when the HIP library or GPU is missing)."""
    import bds_amd
    from bds_amd import native

    assert native.has_test_hooks(), "the GPU tests need the test-hooks build of the library (./build.sh builds it; BDS_LIB_PATH=%s)" % os.environ.get("BDS_LIB_PATH")
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
