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


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_SC3G_FUZZ_N", "32"))))
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


# ---- site relabelling (dnm_subspace.site_perm) ------------------------------------------------------------

def test_relabelled_layout_round_trip(small_layout):
    """A vector in a relabelled layout: reference order -> layout -> reference order is the identity, single
    positions agree with the bulk copy and with the definition (permute the state's bits, then the layout's tables),
    the seeded random state holds the numbers the reference order would."""
    import torch
    from dynamite_amd import backend
    for L, k, seed in ((11, 5, 0), (13, 6, 1), (14, 3, 2)):
        sub = SpinConserve(L, k)
        n = sub.get_dimension()
        perm = np.random.RandomState(seed).permutation(L).astype(np.int8)
        d = backend.with_site_perm(sub._c(), perm)
        v = backend.Vec(n, swz=sub.vec_swizzle, sub_c=d)
        plain = backend.Vec(n, swz=sub.vec_swizzle, sub_c=sub._c())
        assert v.perm == tuple(int(b) for b in perm) and plain.perm is None
        x = rand_state(n, seed=L)
        v.set_local_from_numpy(x)
        assert np.array_equal(v.local_numpy(), x)
        pos = v.positions(torch.arange(n, device=v.array.device)).cpu().numpy()
        assert len(set(pos.tolist())) == n
        assert np.array_equal(v.array.cpu().numpy()[pos], x)
        # the definition: index -> state -> bits moved (spin i -> bit perm[i]) -> position in the plain layout
        states = orc_sub(sub).i2s(np.arange(n))
        moved = np.zeros_like(states)
        for i in range(L):
            moved |= ((states >> i) & 1) << int(perm[i])
        idx_moved = orc_sub(sub).s2i(moved)
        ppos = plain.positions(torch.from_numpy(idx_moved).to(v.array.device)).cpu().numpy()
        assert np.array_equal(pos, ppos)
        for i in (0, n // 3, n - 1):
            assert v.positions(int(i)) == pos[i]
        v.set_random(7)
        plain.set_random(7)
        assert np.array_equal(v.local_numpy(), plain.local_numpy())
        # vectors of different layouts combine through the reference order
        assert abs(v.dot(plain) - plain.dot(plain)) < 1e-12 * n
        w = plain.copy()
        w.axpby(2.0, 1.0, v)
        assert np.abs(w.local_numpy() - 3.0 * plain.local_numpy()).max() < 1e-13
        w2 = backend.Vec(n, swz=sub.vec_swizzle, sub_c=d)
        plain.copy(w2)
        assert np.array_equal(w2.local_numpy(), plain.local_numpy())


@pytest.mark.parametrize("mode", ["auto", "random", "off"])
def test_relabelled_multiply_vs_oracle(small_layout, mode):
    """The same operator in the layout the chooser picks, in a random relabelling and without one: all equal the
    oracle; the chosen relabelling has no more gathered hops than the identity."""
    from dynamite_amd import backend
    L, k = 14, 7
    H = pair_graph(L, seed=77, nbonds=24, complex_hops=True)
    sub = SpinConserve(L, k)
    x = rand_state(sub.get_dimension(), seed=8)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    sp = {"auto": None, "off": False, "random": np.random.RandomState(5).permutation(L).astype(np.int8)}[mode]
    mat = shell(H, sub, site_perm=sp)
    assert (mat.perm_left is None) == (mode == "off")
    if mat.uses_cached_diagonal():
        mat.precompute_diagonal()
    assert np.abs(mult_numpy(mat, x) - want).max() <= tol_for(H, L, x), mat.describe()
    # a state in the subspace's own layout goes through the conversion of ShellMat.mult
    from gpu_util import vec_for
    xv, yv = vec_for(sub), vec_for(sub)
    xv.set_local_from_numpy(x)
    mat.mult(xv, yv)
    assert np.abs(yv.local_numpy() - want).max() <= tol_for(H, L, x)
    # the diagonal is handed out in reference order whatever the layout
    from dynamite_amd import _lib
    if mat.uses_cached_diagonal():
        got = np.empty(sub.get_dimension())
        _lib.check(_lib.lib().dnm_mat_get_diagonal(mat.handle, _lib.pf64(got), backend._stream()))
        assert np.abs(got - orc.precompute_diagonal(orc_msc(H), orc_sub(sub))).max() < 1e-12
    if mode == "auto":
        def gathered(m):
            d = m.describe()
            import re
            return sum(int(v) for v in re.findall(r"(\d+) gathered", d))
        ident = shell(H, sub, site_perm=False)
        assert gathered(mat) <= gathered(ident)
        ident.destroy()
    mat.destroy()


def test_kagome_solvers_in_the_relabelled_layout(small_layout):
    """eigsolve and evolve on the 12-site kagome torus (the flow of run_kagome.py:51-77 without XParity): eigenvalues
    against the reference-built matrix (tests/golden/kagome.npz), eigenvectors and evolved states -- handed back as
    states of the subspace -- against the oracle's multiply."""
    from dynamite_amd.states import State
    g = np.load(os.path.join(GOLDEN, "kagome.npz"))
    H = models.kagome("12")
    sub = SpinConserve(12, 6)
    H.add_subspace(sub)
    mat = H.get_mat(subspaces=(sub, sub))
    assert mat.perm_left is not None and "bond graph" in mat.describe()
    vals, vecs = H.eigsolve(nev=2, getvecs=True, subspace=sub, tol=1e-12)
    assert np.abs(vals[:2] - g["kagome_12_sc/evals_lowest"][:2]).max() < 1e-10
    for lam, v in zip(vals[:1], vecs[:1]):
        xv = v.to_numpy()
        Hx = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xv)
        assert np.linalg.norm(Hx - lam * xv) < 1e-9
        assert abs(v.norm() - 1.0) < 1e-12
    psi = State(L=12, subspace=sub, state='random', seed=3)
    x0 = psi.to_numpy()
    out = H.evolve(psi, t=0.7)
    import scipy.sparse.linalg as spla
    import scipy.sparse as sp
    n = sub.get_dimension()
    cols = np.eye(n, dtype=np.complex128)
    A = np.stack([orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), cols[:, j].copy()) for j in range(n)], axis=1)
    want = spla.expm_multiply(-0.7j * sp.csc_matrix(A), x0)
    got = out.to_numpy()
    assert abs(1 - np.vdot(want, got) / np.vdot(got, got)) < 1e-9
    # evolve made psi adopt the operator's layout: its content is unchanged, later products move nothing
    assert psi.vec.perm == mat.perm_right and out.vec.perm == mat.perm_left
    assert np.array_equal(psi.to_numpy(), x0)
    y = H.dot(psi)
    assert np.abs(y.to_numpy() - A @ x0).max() < 1e-12
    assert abs(psi.dot(out) - np.vdot(out.to_numpy(), x0)) < 1e-12
    fresh = State(L=12, subspace=sub, state='random', seed=3)       # the subspace's own layout
    assert fresh.vec.perm is None and abs(fresh.dot(psi) - 1.0) < 1e-12
    y2 = H.dot(fresh)
    assert fresh.vec.perm == mat.perm_right and np.abs(y2.to_numpy() - A @ x0).max() < 1e-12
    H.destroy_mat()


# ---- XParity on top (the default of run_kagome.py:51-66) ---------------------------------------------------

def _xparity_case(H, L, sector, seed=0):
    """(multiply in the layout's first half, the same through reference-order vectors and the row kernels)"""
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import XParity
    outs = []
    for layout in (True, False):
        config.sc_xparity_layout = layout
        try:
            sub = XParity(SpinConserve(L, L // 2), sector)
            Hc = H.copy()
            Hc.add_subspace(sub)
            x = State(L=L, subspace=sub, state='random', seed=seed)
            assert x.vec.internal == layout
            y = Hc.dot(x)
            outs.append((x.to_numpy(), y.to_numpy(), Hc.get_mat(subspaces=(sub, sub)).describe()))
            Hc.destroy_mat()
        finally:
            config.sc_xparity_layout = True
    return outs


@pytest.mark.parametrize("sector", ['+', '-'])
@pytest.mark.parametrize("kind", ["kagome12", "graph14", "chain14"])
def test_xparity_in_the_layout(small_layout, kind, sector):
    """XParity(SpinConserve(L, L/2)): the hops that touch spin L-1 come composed with the global flip (every spin but
    the pair flipped, subspaces.py:632-674).  In the layout's first half they are gathered hops with the Lo pattern's
    rank taken after the complement (lo pass) or the columns counted from the other end (window pass).  Against the
    reference-order kernels, and for the 12-site kagome torus against the reference's own reduced matrix."""
    if kind == "kagome12":
        H, L = models.kagome("12"), 12
    elif kind == "graph14":
        H, L = pair_graph(14, seed=21, nbonds=30, complex_hops=False, fields=False), 14
    else:
        H, L = models.heisenberg(14), 14
    (x1, y1, d1), (x2, y2, d2) = _xparity_case(H, L, sector)
    assert "bond graph" in d1 and "internal layout" in d1, d1
    assert "internal layout" not in d2
    assert np.array_equal(x1, x2)                 # the seeded state is the same in both layouts
    assert np.abs(y1 - y2).max() <= 1e-12 * max(1.0, np.abs(y2).max()), d1
    if kind == "kagome12":
        from dynamite_amd.states import State
        from dynamite_amd.subspaces import XParity
        g = np.load(os.path.join(GOLDEN, "kagome.npz"))
        pre = "kagome_12_sc_xparity_%s/" % ("plus" if sector == '+' else "minus")
        sub = XParity(SpinConserve(12, 6), sector)
        H.add_subspace(sub)
        x = State(L=12, subspace=sub)
        x.vec.set_local_from_numpy(g[pre + "x"])
        x.set_initialized()
        y = H.dot(x)
        assert np.abs(y.to_numpy() - g[pre + "y"]).max() <= 1e-12
        vals = H.eigsolve(nev=2, subspace=sub, tol=1e-12)
        # (the '-' sector's lowest level is doubly degenerate on this torus: a Krylov space from one start vector holds
        # one copy of it, as SLEPc's would -- the second value returned is the next level)
        want = g[pre + "evals_lowest"]
        assert abs(vals[0] - want[0]) < 1e-10 and np.abs(want - vals[1]).min() < 1e-10
        H.destroy_mat()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_SC3G_FUZZ_X_N", "8"))))
def test_fuzz_xparity_pair_graphs(small_layout, seed):
    rs = np.random.RandomState(300 + seed)
    L = int(rs.choice([12, 14]))
    H = pair_graph(L, seed=50 + seed, nbonds=int(rs.randint(4, 40)), complex_hops=False, fields=False)
    (x1, y1, d1), (x2, y2, d2) = _xparity_case(H, L, '+' if seed & 1 else '-', seed=seed)
    assert "bond graph" in d1, d1
    assert np.abs(y1 - y2).max() <= 1e-12 * max(1.0, np.abs(y2).max()), d1


def test_xparity_production_instance():
    """The flagship case at size: the 30-site kagome torus in XParity(SpinConserve(30, 15)) -- 77.6 M representatives
    in the (14, 10) layout's first half -- against the reference-order kernels on the same seeded state."""
    H = models.kagome("30")
    (x1, y1, d1), (x2, y2, d2) = _xparity_case(H, 30, '-')
    assert "bond graph" in d1 and "[T 6 | W 10 | Lo 14]" in d1, d1
    assert np.array_equal(x1, x2)
    assert np.abs(y1 - y2).max() <= 1e-12 * max(1.0, np.abs(y2).max())
    # ... and the real-arithmetic handle of the same operator against the complex one on a real vector
    import ctypes as C
    import torch
    from dynamite_amd import _lib
    from dynamite_amd.subspaces import XParity
    sub = XParity(SpinConserve(30, 15), '-')
    H.add_subspace(sub)
    cmat, rmat = H.get_mat(subspaces=(sub, sub)), H.get_real_packed_mat(sub)
    assert rmat is not None and rmat.real_packed
    xv, yv = cmat.createVecs()
    xv.set_random(5)
    xv.array.imag.zero_()
    cmat.mult(xv, yv)
    xd = xv.array.real.contiguous()
    yd = torch.empty_like(xd)
    _lib.check(_lib.lib().dnm_mat_mult(rmat.handle, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None))
    torch.cuda.synchronize()
    assert float((yd - yv.array.real).abs().max()) <= 1e-12 * float(yv.array.real.abs().max())
    H.destroy_mat()


# ---- real arithmetic ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("case", ["kagome15", "graph13", "graph14_fields"])
def test_real_packed_graph_multiply(small_layout, case):
    """DNM_MAT_REAL_PACKED on a bond graph: the lo pass on doubles (sc3g_lo_pass_r), the window pass on pairs of
    entries, cached and on-the-fly diagonals, the fused sums of the Lanczos step -- against the oracle."""
    import ctypes as C
    import torch
    from dynamite_amd import _lib
    from gpu_util import vec_for
    if case == "kagome15":
        H, L, k = models.kagome("15"), 15, 7
    elif case == "graph13":
        H, L, k = pair_graph(13, seed=5, nbonds=20, fields=False), 13, 6
    else:
        H, L, k = pair_graph(14, seed=6, nbonds=28, fields=True), 14, 5
    sub = SpinConserve(L, k)
    n = sub.get_dimension()
    mat = shell(H, sub, flags=_lib.MAT_REAL_PACKED, site_perm=False)      # raw vectors of the subspace's own layout below
    assert "bond graph" in mat.describe()
    if mat.uses_cached_diagonal():
        mat.precompute_diagonal()
    xr = np.random.RandomState(L).standard_normal(n)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xr.astype(np.complex128))
    assert np.abs(want.imag).max() == 0.0
    v = vec_for(sub)
    v.set_local_from_numpy(xr.astype(np.complex128))
    xd = v.array.real.contiguous()
    yd = torch.full_like(xd, 7.0)
    _lib.check(_lib.lib().dnm_mat_mult(mat.handle, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None))
    out = vec_for(sub)
    _lib.check(_lib.lib().dnm_vec_layout_unpack_real(C.byref(sub._c()), None, out.ptr, C.c_void_p(yd.data_ptr()), None))
    got = out.local_numpy()
    assert np.abs(got.imag).max() == 0.0
    assert np.abs(got.real - want.real).max() <= tol_for(H, L, xr), mat.describe()
    d = (C.c_double * 3)()
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None,
                                               0.0, d, None))
    assert abs(d[0] - xr @ want.real) <= 1e-11 * max(1.0, abs(xr @ want.real))
    assert abs(d[2] - want.real @ want.real) <= 1e-11 * max(1e-300, want.real @ want.real)
    mat.destroy()


@pytest.mark.parametrize("arith", ["complex", "real"])
@pytest.mark.parametrize("case", ["no_table_small", "dense_lo_25"] +
                         # (opt-in: 11 s of oracle per case)
                         (["no_table_27b"] if os.environ.get("DNM_TEST_LARGEST") == "1" else []))
def test_lo_hops_by_rank_tables(small_layout, monkeypatch, case, arith):
    """The lo pass's LDS hops WITHOUT the partner table (Sc3Op::ptab): the two-table rank of the flipped pattern -- what an
    operator with more than 32 hops inside Lo runs (dense_lo_25: all 91 pairs of the 14 Lo spins of the (14, 10) instance
    plus a chain through the rest, identity labelling) and what DNM_SC3G_PTAB=0 forces (a random graph on the small
    instance, the 27-site kagome torus on the production one) -- complex and real arithmetic, against the oracle."""
    import ctypes as C
    import torch
    from dynamite_amd import _lib
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum
    from gpu_util import vec_for
    saved = (config.sc_layout, config.sc_layout_min_dim)
    try:
        if case == "no_table_small":
            monkeypatch.setenv("DNM_SC3G_PTAB", "0")
            H, L, k = pair_graph(14, seed=31, nbonds=30, complex_hops=False, fields=True), 14, 6
        else:
            config.sc_layout, config.sc_layout_min_dim = (14, 10), 0
            if case == "no_table_27b":
                monkeypatch.setenv("DNM_SC3G_PTAB", "0")
                H = models.kagome("27b")
                L, k = H.L, 13
            else:
                L, k = 25, 12
                rs = np.random.RandomState(9)
                pairs = [(i, j) for i in range(14) for j in range(i + 1, 14)] + [(i, i + 1) for i in range(13, L - 1)]
                H = op_sum(float(rs.uniform(-1, 1)) * (sigmax(i) * sigmax(j) + sigmay(i) * sigmay(j)) +
                           float(rs.uniform(-1, 1)) * sigmaz(i) * sigmaz(j) for i, j in pairs)
                H.L = L
        sub = SpinConserve(L, k)
        n = sub.get_dimension()
        mat = shell(H, sub, flags=_lib.MAT_REAL_PACKED if arith == "real" else 0, site_perm=False)
        d = mat.describe()
        assert "bond graph" in d, d
        if case == "dense_lo_25":
            assert "91 hops in LDS" in d, d
        if mat.uses_cached_diagonal():
            mat.precompute_diagonal()
        xr = np.random.RandomState(L + 1).standard_normal(n)
        x = xr.astype(np.complex128) if arith == "real" else xr + 1j * np.random.RandomState(L + 2).standard_normal(n)
        want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=4)
        if arith == "complex":
            got = mult_numpy(mat, x)
            assert np.abs(got - want).max() <= tol_for(H, L, x) * (4 if case == "dense_lo_25" else 1), d
        else:
            v = vec_for(sub)
            v.set_local_from_numpy(x)
            xd = v.array.real.contiguous()
            yd = torch.full_like(xd, 7.0)
            _lib.check(_lib.lib().dnm_mat_mult(mat.handle, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None))
            out = vec_for(sub)
            _lib.check(_lib.lib().dnm_vec_layout_unpack_real(C.byref(sub._c()), None, out.ptr, C.c_void_p(yd.data_ptr()), None))
            got = out.local_numpy()
            assert np.abs(got.imag).max() == 0.0
            assert np.abs(got.real - want.real).max() <= tol_for(H, L, xr) * (4 if case == "dense_lo_25" else 1), d
        mat.destroy()
    finally:
        config.sc_layout, config.sc_layout_min_dim = saved


@pytest.mark.parametrize("mode", ["basis_free", "restarted"])
def test_kagome_eigsolve_real_arithmetic(small_layout, monkeypatch, mode):
    """eigsolve of the 12-site kagome torus in real arithmetic, in the relabelled layout: eigenvalues against the
    reference-built matrix, the eigenvector -- handed back as a complex state of the subspace -- against the oracle."""
    g = np.load(os.path.join(GOLDEN, "kagome.npz"))
    H = models.kagome("12")
    sub = SpinConserve(12, 6)
    H.add_subspace(sub)
    config.eigs_real_arithmetic = True
    monkeypatch.setenv("DNM_EIGS_BASISFREE", "1" if mode == "basis_free" else "0")
    try:
        from dynamite_amd.computations import eigsolve
        vals, vecs = H.eigsolve(nev=1, getvecs=True, subspace=sub, tol=1e-11)
        assert eigsolve.last_stats["real_arithmetic"]
    finally:
        config.eigs_real_arithmetic = None
    assert abs(vals[0] - g["kagome_12_sc/evals_lowest"][0]) < 1e-9
    xv = vecs[0].to_numpy()
    Hx = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xv)
    assert np.linalg.norm(Hx - vals[0] * xv) < 1e-8 and abs(np.linalg.norm(xv) - 1.0) < 1e-12
    H.destroy_mat()


@pytest.mark.parametrize("sector", ['+', '-'])
@pytest.mark.parametrize("kind", ["kagome12", "graph14", "chain14"])
def test_xparity_real_arithmetic(small_layout, kind, sector):
    """XParity(SpinConserve) in REAL arithmetic: the lo pass ranks the complemented pattern as in complex128, the window
    pass (on pairs of entries) takes the pairs apart where the flip-composed hops read columns from the other end of the
    row.  The real handle's multiply against the complex one on the same real vector, then eigsolve against the
    reference-built spectrum (kagome-12) / the complex solve."""
    import ctypes as C
    import torch
    from dynamite_amd import _lib
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import XParity
    from dynamite_amd.computations import eigsolve
    if kind == "kagome12":
        H, L = models.kagome("12"), 12
    elif kind == "graph14":
        H, L = pair_graph(14, seed=23, nbonds=30, complex_hops=False, fields=False), 14
    else:
        H, L = models.heisenberg(14), 14
    sub = XParity(SpinConserve(L, L // 2), sector)
    H.add_subspace(sub)
    cmat = H.get_mat(subspaces=(sub, sub))
    rmat = H.get_real_packed_mat(sub)
    assert rmat is not None and rmat.real_packed and "bond graph" in rmat.describe()
    assert rmat.perm_left == cmat.perm_left
    xv, yv = cmat.createVecs()
    xv.set_random(11)
    xv.array.imag.zero_()
    cmat.mult(xv, yv)
    xd = xv.array.real.contiguous()
    yd = torch.full_like(xd, 3.0)
    _lib.check(_lib.lib().dnm_mat_mult(rmat.handle, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None))
    torch.cuda.synchronize()
    want = yv.array.real
    assert float((yd - want).abs().max()) <= 1e-12 * max(1.0, float(want.abs().max())), rmat.describe()
    assert float(yv.array.imag.abs().max()) == 0.0
    config.eigs_real_arithmetic = True
    try:
        er = H.eigsolve(nev=1, subspace=sub, tol=1e-11)
        assert eigsolve.last_stats["real_arithmetic"]
        config.eigs_real_arithmetic = False
        ec = H.eigsolve(nev=1, subspace=sub, tol=1e-11)
        assert not eigsolve.last_stats["real_arithmetic"]
    finally:
        config.eigs_real_arithmetic = None
    assert abs(er[0] - ec[0]) < 1e-9
    if kind == "kagome12":
        g = np.load(os.path.join(GOLDEN, "kagome.npz"))
        pre = "kagome_12_sc_xparity_%s/" % ("plus" if sector == '+' else "minus")
        assert abs(er[0] - g[pre + "evals_lowest"][0]) < 1e-9
    H.destroy_mat()


def test_states_in_a_relabelled_layout_behave_like_any_state(small_layout, tmp_path):
    """A state that has adopted an operator's relabelled layout (an eigenvector of the kagome model) through the rest of
    the State surface: reduced density matrix and entropy against the oracle's, save / from_file, set_product, project,
    expectation -- everything index-wise stays in the reference's order (states.py:102-241, 403-447, 627-701)."""
    from dynamite_amd.states import State
    from dynamite_amd.computations import reduced_density_matrix, entanglement_entropy
    from dynamite_amd.operators import sigmaz
    H = models.kagome("15")
    sub = SpinConserve(15, 7)
    H.add_subspace(sub)
    vals, vecs = H.eigsolve(nev=1, getvecs=True, subspace=sub, tol=1e-11)
    psi = vecs[0]
    assert psi.vec.perm is not None
    x = psi.to_numpy()
    keep = [0, 3, 4, 9]
    rho = reduced_density_matrix(psi, keep)
    want = orc.rdm(orc_sub(sub), x, np.array(keep, dtype=np.int64))
    assert np.abs(rho - want).max() < 1e-12
    s = entanglement_entropy(psi, keep)
    w = np.linalg.eigvalsh(want)
    assert abs(s + np.sum(w[w > 1e-300] * np.log(w[w > 1e-300]))) < 1e-10
    fn = str(tmp_path / "kagome_state")
    psi.save(fn)
    back = State.from_file(fn)
    assert np.array_equal(back.to_numpy(), x) and back.vec.perm is None
    assert abs(back.dot(psi) - 1.0) < 1e-12                                  # two layouts, one inner product
    Z = sigmaz(3)
    Z.L = 15
    Z.add_subspace(sub)
    ez = Z.expectation(psi)
    st = orc_sub(sub).i2s(np.arange(sub.get_dimension()))
    assert abs(ez - np.sum((1 - 2 * ((st >> 3) & 1)) * np.abs(x) ** 2)) < 1e-12
    # an index-wise write into a vector of the relabelled layout
    phi = State(L=15, subspace=sub)
    phi._vec = H.get_mat(subspaces=(sub, sub)).createVecs()[0]
    phi.set_product('U' * 8 + 'D' * 7)
    arr = phi.to_numpy()
    idx = sub.state_to_idx(int('1' * 7 + '0' * 8, 2))
    assert arr[idx] == 1.0 and np.count_nonzero(arr) == 1
    e1 = H.expectation(phi)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), arr)
    assert abs(e1 - np.vdot(arr, ref).real) < 1e-12
    H.destroy_mat()
    Z.destroy_mat()


def test_set_product_on_an_adopted_xparity_state(small_layout):
    """An XParity(SpinConserve) state that has adopted an operator's relabelled layout (the layout's first half):
    index-wise writes through Vec.positions(int) -- State.set_product -- land where the reference order says
    (dnm_vec_layout_positions_host used to refuse every relabelled vector that is not the whole layout: ADVICE r5)."""
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import XParity
    H = models.kagome("12")
    N = H.L
    parent = SpinConserve(N, N // 2)
    sub = XParity(parent, sector='+')
    H.add_subspace(sub)
    psi = State(L=N, subspace=sub, state='random', seed=5)
    phi = H.dot(psi)                          # psi adopts the operator's layout (Operator.dot)
    mat = H.get_mat(subspaces=(sub, sub))
    if psi.vec.perm is None:
        pytest.skip("this build keeps the identity labelling for the 12-site torus")
    assert psi.vec.half and phi.vec.perm == psi.vec.perm
    before = psi.to_numpy().copy()
    # positions of single indices agree with the device path's
    import torch
    some = [0, 1, 17, sub.get_dimension() - 1]
    dev = psi.vec.positions(torch.tensor(some, dtype=torch.int64)).cpu().numpy()
    assert [psi.vec.positions(i) for i in some] == dev.tolist()
    # a product state written index-wise into the adopted vector: a representative (spin N-1 up)
    s = 'U' * (N // 2 - 1) + 'D' * (N // 2) + 'U'
    psi.set_product(s)
    arr = psi.to_numpy()
    assert np.count_nonzero(arr) == 1 and arr[sub.state_to_idx(State.str_to_state(s, N))] == 1.0
    assert np.count_nonzero(before) > 1
    H.destroy_mat()
    del mat
