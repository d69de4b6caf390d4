import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def rel_rms(a, b):
    """RMS of (a-b) relative to the RMS of b -- the north_star's 1e-5 metric."""
    import numpy as np

    a = np.asarray(a).astype(np.complex128)
    b = np.asarray(b).astype(np.complex128)
    den = np.sqrt(np.mean(np.abs(b) ** 2))
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2)) / (den if den > 0 else 1.0))
