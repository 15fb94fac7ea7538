import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size parity cases (tens of seconds of CPU oracle time)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 in fp64."""
    import torch
    a = torch.as_tensor(a).double().flatten().cpu()
    b = torch.as_tensor(b).double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
