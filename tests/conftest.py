import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import hf_oracle
    hf_oracle.build()
    return hf_oracle


@pytest.fixture(scope="session")
def hf():
    """the product package; GPU tests fail loudly if libhf.so or the device is missing"""
    import torch
    import hf_amd
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    hf_amd._capi.lib()
    return hf_amd
