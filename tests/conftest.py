import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The host process owns its environment (the library no longer sets anything on its own): eight hardware queues for the
# library's streams, before anything initialises the HIP runtime; and the test-only fault injection of the contexts created
# in this process and its children (nrq_ctx_set_option "fail_after": tests/test_gpu_faults.py).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("NANORQ_HIP_FAULT_INJECT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


_TORCH_UP = False


def pytest_runtest_setup(item):
    """Before the first GPU test touches the HIP path: bring torch's GPU runtime up.  torch ships its own copy of the
    HIP runtime and only finds the device if it initialises BEFORE the runtime libnanorq_hip.so links against opens
    it (bench.py has the same order); some GPU tests keep blocks in HBM as torch tensors."""
    global _TORCH_UP
    if _TORCH_UP or item.get_closest_marker("gpu") is None:
        return
    _TORCH_UP = True
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.empty(1, device="cuda")
    except ImportError:
        pass
