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
    """The CPU checker (oracle/liboracle.so); built on demand with gcc."""
    from oracle import capi
    capi.lib()
    return capi


@pytest.fixture(scope="session")
def gpu():
    """The product on cuda:0.  Fails loudly if the HIP library is missing or no GPU is visible."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import imagestitch_amd
    imagestitch_amd.load()
    return imagestitch_amd
