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
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


# Constants shared by the fixtures (tests/golden/make_golden.py) and the tests.
STATS = dict(velocity_mean=[1.5e-4, -2.5e-4, 0.5e-4], velocity_std=[2.1e-3, 3.2e-3, 1.9e-3],
             acceleration_mean=[1.0e-6, -8.0e-6, 2.0e-6], acceleration_std=[2.4e-4, 3.1e-4, 2.2e-4])
BOUNDS = dict(lower_bounds=[0.1, 0.1, 0.1], upper_bounds=[0.9, 0.9, 0.9])
CART, MAT, CTRL = [2, 3, 4], [1], [5, 6, 7]
G1_CASES = ["dense200", "mixed500", "sparse64", "mean20_3000", "cap5", "cap40_r03"]
