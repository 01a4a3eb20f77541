"""
SpinConserve multiply for operators on ANY bond graph in the three-field internal layout (dynamite_amd/csrc/
sc3g_kernels.hip): the kagome Heisenberg model of the reference's flagship example (examples/scripts/kagome/
run_kagome.py:20-77), random pair-exchange graphs with complex, direction-dependent hops, chains forced through the
graph kernels -- against the oracle's MatMult_CPU_General restatement (bpetsc_template_2.c:371-412, index maps
bsubspace_impl.h:187-245) and the reference-generated fixtures of tests/golden/kagome.npz.
Small sizes run the (a, w) = (6, 4) kernel instances; the production instances (14, 10) run from L = 25 on.
"""
import os

import numpy as np
import pytest

from dynamite_amd import models, lattices
from dynamite_amd.config import config
from dynamite_amd.subspaces import SpinConserve
from gpu_util import shell, orc_msc, orc_sub, rand_state, mult_numpy
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture
def small_layout():
    old = (config.sc_layout, config.sc_layout_min_dim)
    config.sc_layout, config.sc_layout_min_dim = (6, 4), 0
    yield
    config.sc_layout, config.sc_layout_min_dim = old


def pair_graph(L, seed, nbonds=None, complex_hops=False, fields=True):
    """Random bond graph: exchange J (XX + YY) + Jz ZZ on random pairs (any distance), optionally a
    Dzyaloshinskii-Moriya part D (XY - YX) -- matrix elements that differ in the two directions and are complex --
    and random fields."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum
    rs = np.random.RandomState(seed)
    pairs = [(i, j) for i in range(L) for j in range(i + 1, L)]
    pick = rs.choice(len(pairs), size=min(len(pairs), nbonds or 2 * L), replace=False)
    terms = []
    for q in pick:
        i, j = pairs[q]
        J, Jz, D = rs.uniform(-1, 1, 3)
        terms.append(J * (sigmax(i) * sigmax(j) + sigmay(i) * sigmay(j)) + Jz * sigmaz(i) * sigmaz(j))
        if complex_hops:
            terms.append(D * (sigmax(i) * sigmay(j) - sigmay(i) * sigmax(j)))
    if fields:
        terms += [rs.uniform(-1, 1) * sigmaz(i) for i in range(L)]
    H = op_sum(terms)
    H.L = L
    return H


def tol_for(H, L, x):
    return 64 * 2.2e-16 * (2 * L + 2) * max(1.0, np.abs(H.msc['coeffs']).max()) * np.abs(x).max()


@pytest.mark.parametrize("L,k", [(11, 5), (12, 6), (13, 4), (14, 7), (15, 7)])
@pytest.mark.parametrize("kind", ["real", "complex"])
def test_graph_multiply_vs_oracle(small_layout, monkeypatch, kind, L, k):
    H = pair_graph(L, seed=10 * L + k, complex_hops=kind == "complex")
    sub = SpinConserve(L, k)
    x = rand_state(sub.get_dimension(), seed=3)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    for env in ({}, {"DNM_SC3_DIAG": "cached"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        mat = shell(H, sub)
        d = mat.describe()
        assert "bond graph" in d and "internal layout" in d, d
        assert ("real symmetric" in d) == (kind == "real")
        if mat.uses_cached_diagonal():
            mat.precompute_diagonal()
        got = mult_numpy(mat, x)
        assert np.abs(got - want).max() <= tol_for(H, L, x), (kind, env, d)
        mat.destroy()
        for k_ in env:
            monkeypatch.delenv(k_)


@pytest.mark.parametrize("name", ["12", "15", "18a"])
def test_kagome_vs_oracle(small_layout, name):
    """The reference's kagome clusters in SpinConserve(N, N // 2): bond-graph passes against the oracle, and for the
    12- and 15-site tori against y = H x of the reference's own matrix builder (tests/golden/kagome.npz)."""
    H = models.kagome(name)
    N = H.L
    sub = SpinConserve(N, N // 2)
    mat = shell(H, sub)
    assert "bond graph" in mat.describe(), mat.describe()
    if mat.uses_cached_diagonal():
        mat.precompute_diagonal()
    x = rand_state(sub.get_dimension(), seed=1)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    got = mult_numpy(mat, x)
    assert np.abs(got - want).max() <= tol_for(H, N, x)
    if name in ("12", "15"):
        g = np.load(os.path.join(GOLDEN, "kagome.npz"))
        pre = "kagome_%s_sc/" % name
        H.reduce_msc()
        from gpu_util import marshal
        masks, offs, signs, coeffs = marshal(H)
        assert np.array_equal(masks, g[pre + "masks"]) and np.array_equal(signs, g[pre + "signs"])
        assert np.array_equal(coeffs, g[pre + "coeffs"])
        got = mult_numpy(mat, g[pre + "x"])
        assert np.abs(got - g[pre + "y"]).max() <= 1e-12
    mat.destroy()


def test_chain_through_graph_kernels(small_layout, monkeypatch):
    """DNM_SC3_GRAPH=1 sends a chain through the bond-graph passes: the same result as the chain kernels."""
    L, k = 14, 6
    sub = SpinConserve(L, k)
    x = rand_state(sub.get_dimension(), seed=5)
    for H in (models.mbl(L), models.xxz(L)):
        ref = shell(H, sub)
        assert "bond graph" not in ref.describe()
        want = mult_numpy(ref, x)
        ref.destroy()
        monkeypatch.setenv("DNM_SC3_GRAPH", "1")
        mat = shell(H, sub)
        monkeypatch.delenv("DNM_SC3_GRAPH")
        assert "bond graph" in mat.describe()
        got = mult_numpy(mat, x)
        assert np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max())
        mat.destroy()


def test_fields_that_leave_the_subspace_are_skipped(small_layout):
    """The harness's long-range model (benchmarking/benchmark.py:139-146) has single-spin X and Y fields: inside
    SpinConserve they never act, and the operator is a chain with a long-range diagonal -- the tiled passes."""
    L, k = 13, 6
    H = models.bench_long_range(L)
    sub = SpinConserve(L, k)
    mat = shell(H, sub)
    d = mat.describe()
    assert "two-pass" in d and "diagonal cached" in d, d
    mat.precompute_diagonal()
    x = rand_state(sub.get_dimension(), seed=2)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    assert np.abs(mult_numpy(mat, x) - want).max() <= tol_for(H, L, x)
    mat.destroy()


@pytest.mark.parametrize("seed", range(32))
def test_fuzz_pair_graphs(small_layout, seed):
    """Random sizes, fillings, graphs (sparse to all-to-all), real and complex hops."""
    rs = np.random.RandomState(1000 + seed)
    L = int(rs.randint(11, 16))
    k = int(rs.randint(1, L))
    nb = int(rs.randint(1, L * (L - 1) // 2 + 1))
    H = pair_graph(L, seed=seed, nbonds=nb, complex_hops=bool(seed & 1), fields=bool(seed & 2))
    sub = SpinConserve(L, k)
    mat = shell(H, sub)
    d = mat.describe()
    if mat.uses_cached_diagonal():
        mat.precompute_diagonal()
    x = rand_state(sub.get_dimension(), seed=seed)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    got = mult_numpy(mat, x)
    assert np.abs(got - want).max() <= tol_for(H, L, x), (L, k, nb, d)
    mat.destroy()


@pytest.mark.parametrize("case", ["graph25", "kagome27b"])
def test_production_instances_vs_oracle(case):
    """The (14, 10) instances: a random graph at L = 25 and the 27-site kagome torus against the oracle."""
    if case == "graph25":
        H, L, k = pair_graph(25, seed=7, nbonds=44, complex_hops=True), 25, 12
    else:
        H = models.kagome("27b")
        L, k = H.L, 13
    sub = SpinConserve(L, k)
    assert sub.vec_swizzle == (14 | (10 << 8))
    mat = shell(H, sub)
    assert "bond graph" in mat.describe(), mat.describe()
    if mat.uses_cached_diagonal():
        mat.precompute_diagonal()
    x = rand_state(sub.get_dimension(), seed=4)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    got = mult_numpy(mat, x)
    assert np.abs(got - want).max() <= tol_for(H, L, x)
    mat.destroy()
