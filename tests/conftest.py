import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _inference_by_default():
    """The render path is inference: run every test with autograd off, as the reference's eval loop does, so the
    fields take the fused kernels.  Training tests switch it back on with ``torch.enable_grad()``."""
    import torch
    with torch.no_grad():
        yield


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from quadraturefields_amd import _C, build
    if not os.path.exists(_C.LIB_PATH):
        build.build()
    return _C.lib()


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
