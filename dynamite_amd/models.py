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


def bond_heisenberg(edges, L=None, J=1.0):
    """sum over the bonds (i, j) of J * 0.25 (XX + YY + ZZ): the Heisenberg model on any bond graph
    (examples/scripts/kagome/run_kagome.py:12-28 in the reference tree)."""
    H = op_sum(op_sum(J * 0.25 * s(i) * s(j) for s in (sigmax, sigmay, sigmaz)) for i, j in edges)
    if L is not None:
        H.L = L
    return H


def kagome(cluster):
    """Nearest-neighbour Heisenberg model on a kagome torus of the reference's cluster library
    (run_kagome.py:20-28; clusters and vertex numbering: ``lattices.kagome``)."""
    from . import lattices
    n, edges = lattices.kagome(cluster)
    return bond_heisenberg(edges, L=n)


def bench_long_range(L):
    """benchmarking/benchmark.py:139-146 ('long_range': ZZ between all pairs, nearest-neighbour XX, small fields)."""
    H = op_sum(index_sum(0.25 * sigmaz(0) * sigmaz(i), size=L) for i in range(1, L))
    H += 0.5 * index_sum(0.25 * sigmax(0) * sigmax(1), size=L)
    H += op_sum(0.05 * index_sum(s(), size=L) for s in (sigmax, sigmay, sigmaz))
    H.L = L
    return H


def bench_ising(L):
    """benchmarking/benchmark.py:162-163."""
    H = index_sum(0.25 * sigmaz(0) * sigmaz(1), size=L) + 0.1 * index_sum(sigmax(), size=L)
    H.L = L
    return H


def bench_xx(L):
    """benchmarking/benchmark.py:165-166."""
    H = index_sum(0.25 * sigmax(0) * sigmax(1), size=L)
    H.L = L
    return H


def bench_syk(L):
    """benchmarking/benchmark.py:148-160 (stdlib random.seed(0), uniform(-1, 1) per product, overall scale)."""
    from random import seed, uniform
    seed(0)
    maj = [majorana(i) for i in range(L * 2)]

    def gen():
        for idxs in combinations(range(L * 2), 4):
            p = op_product(maj[i] for i in idxs)
            p.scale(uniform(-1, 1))
            yield p
    H = op_sum(gen())
    H.scale(np.sqrt(6 / (L * 2) ** 3))
    H.L = L
    return H


BY_NAME = {'bench_long_range': bench_long_range, 'bench_ising': bench_ising, 'bench_xx': bench_xx,
           'bench_syk': bench_syk,
           'mbl': mbl, 'heisenberg': heisenberg, 'xxz': xxz, 'ising': ising,
           'long_range': long_range, 'localized': localized, 'syk': syk, 'xsum': xsum}
