import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.build()
    return oracle_lib


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def cube40():
    return _load("cube40_62.npz")


@pytest.fixture(scope="session")
def twocube():
    return _load("twocube10.npz")


@pytest.fixture(scope="session", params=["synth_twosphere_24.npz", "synth_sphere_40x33x27.npz"])
def synth(request):
    return _load(request.param)


def F(a):
    """fresh Fortran-ordered writable copy"""
    return np.array(a, dtype=a.dtype, order="F", copy=True)


def sha(a):
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a.ravel(order="F")).tobytes()).hexdigest()
