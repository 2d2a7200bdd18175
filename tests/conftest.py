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


@pytest.fixture(autouse=True)
def _reset_kernel_variants():
    """Tests may force the tile stage's large-input fallback paths (bds_set_option); restore the defaults afterwards."""
    yield
    try:
        from bilateral_driving_amd import _lib
        if _lib._lib is not None:
            _lib.set_option(_lib.OPT_SHORT_SORT, 1)
            _lib.set_option(_lib.OPT_PACKED, 1)
            _lib.set_option(_lib.OPT_DEBUG, 0)
            _lib.set_option(_lib.OPT_CELLS, 3)
            _lib.set_option(_lib.OPT_SCHED_BINS, 1)
    except Exception:
        pass
