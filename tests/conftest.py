import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def track_path():
    return lambda name: os.path.join(ROOT, "fsae-mpc_amd", "tracks", name + ".json")


@pytest.fixture(scope="session")
def otrack(orc, track_path):
    return orc.Track.load(track_path("fsg2019"))


def golden_files():
    """LTV-MPC instance fixtures (inputs, QP, certified solution; tests/harness/make_golden.py)."""
    d = os.path.join(ROOT, "tests", "golden")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".npz") and not f.startswith("regress_"))


def regress_files():
    """Single QPs an earlier build missed (H, g, A, bounds in the device layout + the certified solution)."""
    d = os.path.join(ROOT, "tests", "golden")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".npz") and f.startswith("regress_"))


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), "finite pattern differs"
    assert np.array_equal(a[~fin], b[~fin]), "infinite entries differ"
    if not fin.any():
        return 0.0
    return float(np.max(np.abs(a[fin] - b[fin])) / max(1.0, float(np.max(np.abs(b[fin])))))
