import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy view of one committed .npz fixture (data only: inputs and expected outputs)."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.index = [str(k) for k in self._z["index"]]

    def get(self, key, field, default=None):
        k = key + "_" + field
        return self._z[k] if k in self._z.files else default


@pytest.fixture(scope="session")
def g1():
    return Golden("g1_tpack.npz")


@pytest.fixture(scope="session")
def g3():
    return Golden("g3_conv.npz")


@pytest.fixture(scope="session")
def g4():
    return Golden("g4_module.npz")


def conv_tolerance(got, exact64, chain32):
    """SURVEY.md section 7 parity rule: the build may deviate from the float64-exact result by
    1e-5 abs, or by as much as the reference's own fp32 accumulation chain does, whichever is larger."""
    err = np.abs(got.astype(np.float64) - exact64)
    allowed = np.maximum(1e-5, np.abs(chain32.astype(np.float64) - exact64))
    return err, allowed
