import json
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


class Golden:
    """Lazy view of tests/golden/*.npz as {case: {key: array}}."""

    def __init__(self, fname):
        self._z = np.load(os.path.join(GOLDEN, fname))
        self.cases = {}
        for k in self._z.files:
            c, key = k.split("/", 1)
            self.cases.setdefault(c, []).append(key)

    def __getitem__(self, case):
        return {k: self._z[case + "/" + k] for k in self.cases[case]}

    def names(self):
        return sorted(self.cases)


@pytest.fixture(scope="session")
def golden_full():
    return Golden("full_space.npz")


@pytest.fixture(scope="session")
def golden_sub():
    return Golden("subspaces.npz")


@pytest.fixture(scope="session")
def golden_xp():
    return Golden("xparity.npz")


@pytest.fixture(scope="session")
def known():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


def cplx(v):
    """JSON complex: number or [re, im]."""
    if isinstance(v, list):
        return complex(v[0], v[1])
    return complex(v)


def cmatrix(rows):
    return np.array([[cplx(v) for v in r] for r in rows], dtype=np.complex128)
