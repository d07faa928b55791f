import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- FIRST: the PyTorch ROCm wheel bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1;
#                 whichever HIP runtime is loaded first serves the whole process (same SONAME), and torch finds no
#                 GPU when it is handed the system runtime that libtwxhip.so would otherwise pull in

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def golden_xval():
    """Cross-validation goldens (tests/golden/make_golden_xval.py; same seeded inputs as ``golden``)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_xval_v1.npz"))


@pytest.fixture(scope="session")
def golden_case(golden):
    """The seeded inputs the goldens were generated on (hash-checked)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    grid, tmin, tmax = make_golden.case_inputs()
    assert make_golden.input_hash(grid, tmin, tmax) == str(golden["input_hash"]), \
        "synthetic generator drifted: regenerate tests/golden (make_golden.py)"
    return grid, tmin, tmax


@pytest.fixture(scope="session")
def orc():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
