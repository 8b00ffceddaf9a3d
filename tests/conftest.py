import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the reference captured by tests/golden/make_golden.py."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")))


@pytest.fixture
def tuning():
    """Setter for the library's tuning table (hpcs_debug_set_tuning: the HPCS_* switches are read from the environment once,
    so tests that A/B a kernel variant replace the value in place); the touched entries are put back afterwards."""
    from hpcs_amd import _lib
    defaults = {"HPCS_KNN_KERNEL": 0, "HPCS_SAMPLED_V": 4, "HPCS_SAMPLED_SEG": 0, "HPCS_EDGE_BWD_ATOMIC": 0, "HPCS_GEMM_X3": 1}
    touched = []

    def set_(name, value):
        if name not in defaults:
            raise KeyError(name)
        _lib.set_tuning(name, value)
        touched.append(name)
    yield set_
    for name in touched:
        _lib.set_tuning(name, defaults[name])
