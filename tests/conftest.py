import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The oracle (torch on the CPU + OpenMP C rasteriser) is the slow half of every parity test.  A GPU box shows all of its
# host's cores (256) to a process that may use 16 of them: torch / OpenMP pools sized by os.cpu_count() then spend their
# time being descheduled (measured: the c2 parity test 63 s against 17 s of oracle work).  Pools of at most 16 threads.
_THREADS = str(min(os.cpu_count() or 1, int(os.environ.get("SPLAT_ONE_AMD_TEST_THREADS", "16"))))
os.environ.setdefault("OMP_NUM_THREADS", _THREADS)
os.environ.setdefault("MKL_NUM_THREADS", _THREADS)


def pytest_configure(config):
    import torch
    torch.set_num_threads(int(_THREADS))
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
