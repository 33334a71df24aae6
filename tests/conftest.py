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


def assert_forward_close(out, ref, rel=1e-5, floor=0.0, what=""):
    """The forward parity bar, the same in every test (north_star: 1e-5 relative float32):
      * against the tensor's largest value: max |out - ref| <= rel * max(max |ref|, floor), and
      * per element: |out - ref| <= rel |ref| + rel rms(ref) everywhere, so that a component much smaller than the tensor's maximum
        is still held in relative terms -- down to the tensor's rms, below which float32 itself (the reference's arithmetic)
        carries no relative information through a multi-step network."""
    out = np.asarray(out, np.float64)
    ref = np.asarray(ref, np.float64)
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert np.isfinite(out).all(), what
    if ref.size == 0:
        return
    err = np.abs(out - ref)
    scale = max(float(np.abs(ref).max()), floor)
    assert err.max() <= rel * scale, (what, "max-normalised", float(err.max()), scale)
    rms = max(float(np.sqrt(np.mean(ref ** 2))), floor)
    worst = float((err / (rel * np.abs(ref) + rel * rms)).max()) if rms > 0 else float(err.max() > 0)
    assert worst <= 1.0, (what, "per-element bound", worst)
