import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must fail loudly, not skip, when there is no device or the HIP library is missing
    pass


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "this test needs a HIP device (no CPU fallback exists)"
    return torch.device("cuda:0")
