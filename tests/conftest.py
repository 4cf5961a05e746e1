import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# MIOpen's default find mode benchmarks many solvers the first time it sees a convolution shape (minutes for the U-Net's
# 2048^2 layers on a fresh box); the tests check results, not speed, so they take the heuristic pick.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture()
def golden_taps(golden):
    """Gaussian taps of the interpreter that generated the goldens (tests/golden/weights.npz), as {sigma: taps}."""
    g = golden("weights")
    return {float(k[2:]): g[k] for k in g.files if k.startswith("w_")}


@pytest.fixture()
def oracle_with_golden_taps(golden_taps):
    from oracle import oracle as orc
    old = dict(orc.TAP_OVERRIDE)
    orc.TAP_OVERRIDE.update(golden_taps)
    yield orc
    orc.TAP_OVERRIDE.clear()
    orc.TAP_OVERRIDE.update(old)
