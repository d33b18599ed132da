import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library; GPU tests fail loudly (not skip) if it is missing."""
    from starcop_amd import _lib
    _lib.require_device()
    return _lib.load()


@pytest.fixture(autouse=True)
def _release_device_tensors():
    yield
    try:
        import hip_ops
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        hip_ops._KEEP.clear()
    except Exception:
        pass
