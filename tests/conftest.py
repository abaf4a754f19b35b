import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fx_reg():
    return dict(np.load(os.path.join(GOLDEN, "fx_registration.npz")))


@pytest.fixture(scope="session")
def fx_vg():
    return dict(np.load(os.path.join(GOLDEN, "fx_voxelgrid.npz")))


@pytest.fixture(scope="session")
def fx_small():
    return dict(np.load(os.path.join(GOLDEN, "fx_small.npz")))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def tri6(c):
    c = np.asarray(c)
    return np.stack([c[..., 0, 0], c[..., 0, 1], c[..., 0, 2], c[..., 1, 1], c[..., 1, 2], c[..., 2, 2]], axis=-1)
