"""
The Hamiltonians of the reference's benchmark harness and integration tests,
built with this package's operator algebra: ``benchmarking/benchmark.py:127-176``
and ``tests/integration/hamiltonians.py:25-82`` in the reference tree, plus the
XXZ chain defined in BASELINE.md.
"""
from itertools import combinations

import numpy as np

from .extras import majorana
from .operators import sigmax, sigmay, sigmaz, index_sum, op_sum, op_product


def mbl(L):
    """Random-field Heisenberg chain, benchmark.py:131-137 ('MBL'):
    sum_i 0.25 (XX+YY+ZZ)_{i,i+1} + sum_i 0.5 h_i Z_i, h_i = uniform(-3,3) after
    stdlib random.seed(0)."""
    from random import seed, uniform
    rtn = index_sum(op_sum(0.25 * s(0) * s(1) for s in (sigmax, sigmay, sigmaz)), size=L)
    seed(0)
    for i in range(L):
        rtn += uniform(-3, 3) * 0.5 * sigmaz(i)
    rtn.L = L
    return rtn


def heisenberg(L):
    """benchmark.py:168-169."""
    rtn = index_sum(op_sum(0.25 * s(0) * s(1) for s in (sigmax, sigmay, sigmaz)), size=L)
    rtn.L = L
    return rtn


def xxz(L, delta=0.5):
    """Open XXZ chain of BASELINE.md: 0.25(XX+YY) + 0.25*delta*ZZ."""
    rtn = index_sum(0.25 * sigmax(0) * sigmax(1) + 0.25 * sigmay(0) * sigmay(1)
                    + 0.25 * delta * sigmaz(0) * sigmaz(1), size=L)
    rtn.L = L
    return rtn


def ising(L):
    """hamiltonians.py:25-31."""
    H = index_sum(sigmaz(0) * sigmaz(1), size=L)
    H += 0.5 * index_sum(sigmax(), size=L)
    H.L = L
    return H


def long_range(L):
    """hamiltonians.py:33-52."""
    alpha = 1.13
    H = index_sum(sigmax(0) * sigmax(1), size=L)
    H += op_sum(index_sum(1 / (i ** alpha) * sigmaz(0) * sigmaz(i), size=L) for i in range(1, L))
    H += index_sum(0.5 * sigmax(), L)
    H += index_sum(0.3 * sigmay(), L)
    H += index_sum(0.1 * sigmaz(), L)
    H.L = L
    return H


def localized(L):
    """hamiltonians.py:54-61 (numpy seed 0)."""
    np.random.seed(0)
    H = index_sum(op_sum(s(0) * s(1) for s in (sigmax, sigmay, sigmaz)), size=L)
    H += op_sum(np.random.uniform(-1, 1) * sigmaz(i) for i in range(L))
    H.L = L
    return H


def syk(L):
    """hamiltonians.py:63-82 (numpy seed 0)."""
    np.random.seed(0)
    maj = [majorana(i) for i in range(L * 2)]

    def gen():
        for idxs in combinations(range(L * 2), 4):
            p = op_product(maj[i] for i in idxs)
            p.scale(np.random.uniform(-1, 1))
            yield p
    H = op_sum(gen())
    H.L = L
    return H


def xsum(L):
    """sum_i sigma_x_i: spectrum -L + 2k (test_eigsolve.py:95-123)."""
    H = index_sum(sigmax(), size=L)
    H.L = L
    return H


BY_NAME = {'mbl': mbl, 'heisenberg': heisenberg, 'xxz': xxz, 'ising': ising,
           'long_range': long_range, 'localized': localized, 'syk': syk, 'xsum': xsum}
