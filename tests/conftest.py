import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


os.environ.setdefault("GSA_TEST_HOOKS", "1")      # gsa_debug_inject (include/gsa.h) answers only in a process that carries this


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def hip_library():
    """Path of the built HIP library (built here if missing; hipcc cross-compiles without a GPU)."""
    import subprocess
    from gan_segmentation_amd import _lib
    if not os.path.exists(_lib.HIP_LIBRARY):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    return _lib.HIP_LIBRARY


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch
