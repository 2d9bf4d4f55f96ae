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


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The in-tree libgeoadv.so normally travels with the snapshot; on a clean checkout build it (hipcc
    cross-compiles without a GPU).  There is still no fallback: if the build fails, the tests fail."""
    from geometric_adv_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle.cpu_oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden_nn():
    return np.load(os.path.join(GOLDEN, "nn_distance.npz"))


@pytest.fixture(scope="session")
def golden_nn_nonfinite():
    return np.load(os.path.join(GOLDEN, "nn_distance_nonfinite.npz"))


@pytest.fixture(scope="session")
def golden_emd():
    return np.load(os.path.join(GOLDEN, "approxmatch.npz"))


@pytest.fixture(scope="session")
def golden_grouping():
    return np.load(os.path.join(GOLDEN, "grouping.npz"))


def cloud(seed, b, n):
    rng = np.random.default_rng(seed)
    return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
