import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the suite drives plan / kernel / solver knobs (DNM_TILE_BITS, DNM_SC_BLOCK, DNM_EXCHANGE ...) to cover shapes the
# defaults do not pick at test sizes; the library honours them only under this gate (csrc/dnm_common.h: knob)
os.environ.setdefault("DNM_EXPERIMENTAL", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "default_layout: run with the production vector layout (full-size tests)")


@pytest.fixture(autouse=True)
def _vector_layout(request):
    """GPU tests run their (small) Full / Parity vectors in a swizzled layout whose shift is small enough to
    permute them (DNM_TEST_SWZ, default 6: index bits [6, 8) folded onto bits [4, 6)); the production shift (16)
    only moves amplitudes beyond 2^16.  Tests marked default_layout keep the production value."""
    from dynamite_amd.config import config as dcfg
    old = dcfg.vec_swizzle
    if request.node.get_closest_marker("gpu") and not request.node.get_closest_marker("default_layout"):
        dcfg.vec_swizzle = int(os.environ.get("DNM_TEST_SWZ", "6"))
    yield
    dcfg.vec_swizzle = old


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
