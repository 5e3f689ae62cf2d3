import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as co
    co.lib()   # builds oracle/libppqc_oracle.so on first use (gcc only)
    return co


@pytest.fixture(scope="session")
def qc_mod():
    from pypore_amd import qc
    return qc


@pytest.fixture(autouse=True, scope="module")
def _release_torch_cache(request):
    """A module that used torch for device memory (the 100 GB resident job of test_gpu_fullsize) leaves it in torch's
    caching allocator, out of reach of the library's own hipMalloc in the modules that follow: hand it back."""
    yield
    if "fullsize" in request.module.__name__ and "torch" in sys.modules:
        import gc
        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.empty_cache()
