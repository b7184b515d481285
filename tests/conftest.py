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


def experiments_lib():
    """File name of the experiments build of the library (MIFLOW_BUILD_VARIANT=exp MIFLOW_EXTRA_FLAGS=-DMIFLOW_EXPERIMENTS python -m
    opencv_contrib_amd.build) if it is in the tree, else None.  The release library holds the kernels a release build can reach
    (VERDICT r05 item 8); the tuning variants that lost their A/B -- and the digest tests of their bit-identity -- live there."""
    name = "libmiflow_exp.so"
    return name if os.path.exists(os.path.join(ROOT, "opencv_contrib_amd", name)) else None


@pytest.fixture(scope="session")
def exp_env():
    """Environment of a subprocess that loads the experiments build; skips where that library has not been built."""
    name = experiments_lib()
    if not name:
        pytest.skip("experiments build (libmiflow_exp.so) not in the tree")
    return dict(os.environ, MIFLOW_LIB=name)


def loaded_library_is_experiments_build():
    """True where the library this process loads (MIFLOW_LIB) is the experiments build.  Usable at collection time (no GPU needed); False
    where no library has been built yet."""
    try:
        from opencv_contrib_amd import capi
        return b"+experiments" in (capi.lib().mi_version() or b"")
    except Exception:
        return False
