import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def diag_lib():
    """Route this test's C-ABI calls through the DIAGNOSTIC build (libavddpg_hip_diag.so, -DAVD_DIAG): the only build in which
    the AVD_* environment switches (kernel variants for cross-checks) exist. The shipped library reads no environment."""
    from avddpg_amd import _hip

    with _hip.diag_library() as lib:
        yield lib
