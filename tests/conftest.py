import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from wcsph_amd import _lib
        return _lib.load().wcsph_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both native pieces must exist before any test runs (the HIP library is cross-compiled on CPU boxes)."""
    import __graft_entry__ as g
    g.build()
    # the checker's OpenMP loops are short: hundreds of threads (GPU-box hosts have 256) only add fork/join cost
    import oracle
    oracle.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
