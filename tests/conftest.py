import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o
    _o.build()
    return _o


@pytest.fixture(scope="session")
def gpu_ctx():
    """One ucfp_ctx on device 0. Fails (not skips) when the HIP library or device is missing:
    a GPU test that silently fell back would prove nothing."""
    from ucfp_amd import _lib
    return _lib.default_context(0)


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    torch.cuda.set_device(0)
    return torch
