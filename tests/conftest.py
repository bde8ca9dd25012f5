import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the longer GPU cases (still part of -m gpu)")


@pytest.fixture(scope="session")
def native_lib():
    """The C-ABI library; built on demand so the CPU suite can check symbols."""
    from adrates_amd import _native
    if not os.path.exists(_native.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _native.load()


@pytest.fixture(scope="session")
def gpu_ctx(native_lib):
    from adrates_amd import _native
    return _native.default_context(0)      # one context for the whole suite: the engine's uploads go through it too
