import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~30 s on the 8-core CPU container")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


LAB_LIB = os.path.join(os.path.dirname(__file__), "..", "benchmarks", "lab", "libdvd_hip_lab.so")


@pytest.fixture
def lab(monkeypatch):
    """Route dvd_amd.lib through the LAB build of the library (make -C dvd_amd/csrc lab: product sources + the
    experiment kernels + the DVD_* environment switches) for one test.  The product library reads no environment
    variable, so the fast-vs-fallback and experiment-variant checks run against this build."""
    import ctypes
    if not os.path.exists(LAB_LIB):
        pytest.skip("benchmarks/lab/libdvd_hip_lab.so is not built (make -C dvd_amd/csrc lab)")
    from dvd_amd import lib
    cdll = lib.bind(ctypes.CDLL(os.path.abspath(LAB_LIB)))
    monkeypatch.setattr(lib, "_lib", cdll)
    return cdll
