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
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    """torch + libmiflow on cuda:0; fails loudly if the HIP extension is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    from opencv_contrib_amd import capi
    capi.lib()
    assert capi.lib().mi_device_count() >= 1
    return torch.device("cuda:0")
