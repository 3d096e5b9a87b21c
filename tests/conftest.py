import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Test infrastructure: make sure the in-tree HIP extension and the C oracle are built and current
    (hipcc cross-compiles gfx950 without a GPU).  The product itself never builds on import."""
    from vbq_amd import build
    build.build_hip()
    build.build_oracle()
