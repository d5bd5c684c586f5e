import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN_NPZ = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (test infrastructure; never imported by the product package)."""
    from oracle.oracle import COracle
    return COracle()


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by executing the reference's own Python source (tests/golden/make_golden.py)."""
    return np.load(GOLDEN_NPZ)


@pytest.fixture(scope="session")
def gpu():
    """A MandelbrotDevice on GPU 0.  Fails loudly (no skip, no fallback) when the HIP path is unusable."""
    from distributedmandelbrot_amd import MandelbrotDevice
    dev = MandelbrotDevice(0)
    yield dev
    dev.close()
