import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the reference itself (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def ctx():
    from cora_amd import _lib

    return _lib.get_context()


@pytest.fixture(scope="session")
def model21():
    """Oracle 21cm model with its lookup tables built once per session (~15 s)."""
    from oracle import models

    m = models.Corr21cm()
    m.tables()
    return m
