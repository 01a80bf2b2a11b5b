import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds")


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from splat_one_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libsplat_one_amd.so is not built"
    _lib.load()
    return torch.device("cuda:0")
