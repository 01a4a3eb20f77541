"""
Vector kernels and the native Krylov drivers, through the dynamite-style API
(Operator.evolve / Operator.eigsolve / State), with the reference's own
acceptance criteria: tests/integration/test_evolve.py:34-57 (|1 - <y,y_ref>/|y|^2|
< 1e-9, norm preserved to 1e-9 for real t), test_eigsolve.py:17-88 (Rayleigh
quotient, residual, orthogonality), and golden expm / eigenvalue vectors.
"""
import ctypes as C

import numpy as np
import pytest

from dynamite_amd import _lib, models, backend
from dynamite_amd.computations import MaxIterationsError, ConvergenceError
from dynamite_amd.operators import Operator
from dynamite_amd.states import State, UninitializedError
from dynamite_amd.subspaces import Full, Parity, SpinConserve, XParity, Explicit
from gpu_util import vec_from, rand_state, dense_spectrum

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ vector kernels

def test_vec_kernels_vs_numpy():
    n = 100003
    x, y = rand_state(n, 1), rand_state(n, 2)
    xv, yv = vec_from(x), vec_from(y)
    assert abs(xv.dot(yv) - np.sum(x * np.conj(y))) < 1e-10      # VecDot(x,y) = sum x conj(y)
    assert abs(xv.norm() - np.linalg.norm(x)) < 1e-10
    yv.axpby(0.3 - 0.2j, 1.5j, xv)
    y = (0.3 - 0.2j) * x + 1.5j * y
    assert np.max(np.abs(yv.local_numpy() - y)) < 1e-13
    yv.scale(2 - 1j)
    assert np.max(np.abs(yv.local_numpy() - (2 - 1j) * y)) < 1e-13
    c = xv.copy()
    assert np.array_equal(c.local_numpy(), x)
    c.set(1 + 2j)
    assert np.all(c.local_numpy() == 1 + 2j)


def test_mdot_maxpy_basis_update():
    L = _lib.lib()
    n, nv = 5000, 11
    V = np.stack([rand_state(n, 10 + j) for j in range(nv)])
    w = rand_state(n, 99)
    Vd, wd = vec_from(V.reshape(-1)), vec_from(w)
    h = np.zeros(2 * nv)
    _lib.check(L.dnm_vec_mdot(Vd.ptr, n, nv, wd.ptr, n, _lib.pf64(h), None))
    ref = V.conj() @ w
    assert np.max(np.abs(h.view(complex) - ref)) < 1e-10
    c = rand_state(nv, 5)
    _lib.check(L.dnm_vec_maxpy(wd.ptr, Vd.ptr, n, nv, n, _lib.pf64(c.view(float).copy()), None))
    assert np.max(np.abs(wd.local_numpy() - (w + c @ V))) < 1e-12
    nout = 4
    S = np.stack([rand_state(nv, 50 + o) for o in range(nout)])     # S[o, j] = S(j, o)
    _lib.check(L.dnm_vec_basis_update(Vd.ptr, n, nv, nout, n, _lib.pf64(S.reshape(-1).view(float).copy()), None))
    out = Vd.local_numpy().reshape(nv, n)
    assert np.max(np.abs(out[:nout] - S @ V)) < 1e-11
    assert np.array_equal(out[nout:], V[nout:])
    # both kernels behind it: registers (nout <= 16) and the LDS-staged one (more outputs), odd shapes
    for nv2, nout2, n2 in ((16, 8, 70001), (17, 16, 4099), (33, 20, 3000), (30, 29, 1025), (200, 180, 1500),
                           (700, 20, 333)):       # (the last two: bases too wide for 64 staged rows)
        V2 = np.stack([rand_state(n2, 200 + j) for j in range(nv2)])
        S2 = np.stack([rand_state(nv2, 300 + o) for o in range(nout2)])
        Vd2 = vec_from(V2.reshape(-1))
        _lib.check(L.dnm_vec_basis_update(Vd2.ptr, n2, nv2, nout2, n2, _lib.pf64(S2.reshape(-1).view(float).copy()), None))
        out2 = Vd2.local_numpy().reshape(nv2, n2)
        assert np.max(np.abs(out2[:nout2] - S2 @ V2)) < 1e-10 and np.array_equal(out2[nout2:], V2[nout2:])


def test_device_rng_moments():
    n = 1 << 20
    v = backend.Vec(n)
    v.set_random(1234)
    a = v.local_numpy()
    assert abs(a.real.mean()) < 5e-3 and abs(a.imag.mean()) < 5e-3
    assert abs(a.real.var() - 1) < 1e-2 and abs(a.imag.var() - 1) < 1e-2
    assert abs(np.mean(a.real * a.imag)) < 5e-3
    v2 = backend.Vec(n)
    v2.set_random(1234)
    assert np.array_equal(v2.local_numpy(), a)          # counter-based: reproducible


def test_state_set_random_matches_reference_stream(golden_full):
    g = golden_full["mbl_L12"]
    s = State(L=12, state='random', seed=0)
    assert np.max(np.abs(s.to_numpy() - g["x"])) < 1e-15


def test_state_basics():
    s = State(L=6)
    with pytest.raises(UninitializedError):
        s.norm()
    s.set_product('DUDUDU')
    a = s.to_numpy()
    assert a[0b010101] == 1 and np.count_nonzero(a) == 1
    s.set_product(5)
    assert s.to_numpy()[5] == 1
    sc = State(L=6, subspace=SpinConserve(6, 3), state='UUUDDD')
    assert sc.to_numpy()[sc.subspace.state_to_idx(0b111000)] == 1
    with pytest.raises(ValueError):
        State(L=6, subspace=SpinConserve(6, 3), state='UUDDDD')
    u = State(L=6, state='uniform')
    assert abs(u.norm() - 1) < 1e-14


# ------------------------------------------------------------------ evolve

def _evolve_check(H, x0, t, ref, **kw):
    y = H.evolve(x0, t=t, **kw)
    ynp = y.to_numpy()
    if np.imag(t) == 0:
        assert abs(1 - np.linalg.norm(ynp)) < 1e-9
    ov = np.vdot(ref, ynp) / np.vdot(ref, ref)
    assert abs(1 - ov) < 1e-9, (t, ov)
    return ynp


@pytest.mark.parametrize("name,L,ts", [("mbl", 12, ["1", "5", "0-0.25j"]), ("mbl", 10, ["1", "0.3-0.2j"]),
                                       ("long_range", 8, ["1"]), ("syk", 5, ["1"]), ("ising", 10, ["1"])])
def test_evolve_golden(golden_full, name, L, ts):
    """BASELINE configs[0] (L=12 random-field Heisenberg, evolve(t=1)) and friends
    against scipy expm_multiply on the reference-built matrix."""
    g = golden_full[f"{name}_L{L}"]
    H = models.BY_NAME[name](L)
    x0 = State(L=L, state='random', seed=0)
    for key in ts:
        t = complex(key) if 'j' in key else float(key)
        k = "expm_t=" + key if 'j' not in key else "expm_t=" + ("%g%+gj" % (t.real, t.imag))
        _evolve_check(H, x0, t, g[k])


@pytest.mark.parametrize("name,L,ts", [("mbl", 12, ["1", "5"]), ("mbl", 10, ["1"]), ("long_range", 8, ["1"]),
                                       ("syk", 5, ["1"]), ("ising", 10, ["1"])])
def test_evolve_chebyshev_golden(golden_full, name, L, ts):
    """algo='chebyshev' (Chebyshev expansion on the fused multiply) against the same golden vectors, element-wise
    agreement with the Krylov result, long times (several steps), negative times, tight tolerances."""
    g = golden_full[f"{name}_L{L}"]
    H = models.BY_NAME[name](L)
    x0 = State(L=L, state='random', seed=0)
    for key in ts:
        t = float(key)
        ynp = _evolve_check(H, x0, t, g["expm_t=" + key], algo='chebyshev')
        ref = g["expm_t=" + key]
        assert np.max(np.abs(ynp - ref)) < 1e-9
        tight = H.evolve(x0, t=t, algo='chebyshev', tol=1e-13).to_numpy()
        assert np.max(np.abs(tight - ref)) < 5e-12
    back = H.evolve(H.evolve(x0, t=40.0, algo='chebyshev'), t=-40.0, algo='chebyshev')     # several steps each way
    assert np.max(np.abs(back.to_numpy() - x0.to_numpy())) < 1e-8
    from dynamite_amd.computations import evolve
    assert evolve.last_stats['its'] >= 1 and evolve.last_stats['matvecs'] > 40
    with pytest.raises(ValueError):
        H.evolve(x0, t=1.0 - 0.5j, algo='chebyshev')


def test_evolve_chebyshev_subspaces(monkeypatch):
    """The recurrence on every kernel family: tiled (Full, Parity), SpinConserve row and block kernels, the
    generic kernel (Explicit) -- against the Krylov result."""
    L = 14
    monkeypatch.setenv("DNM_TILE_BITS", "8")
    monkeypatch.setenv("DNM_LOG_ROWS", "2")
    sc = SpinConserve(L, L // 2)
    cases = [("full", Full(L=L), {}), ("parity", Parity('odd', L=L), {}), ("sc", sc, {"DNM_SC_BLOCK": "0"}),
             ("scblock", sc, {"DNM_SC_BLOCK": "10"}),
             ("explicit", Explicit(sc.idx_to_state(np.arange(0, sc.get_dimension(), dtype=np.int64)), L=L), {})]
    for name, sub, env in cases:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        H = models.heisenberg(L)
        H.add_subspace(sub)
        x0 = State(subspace=sub, state='random', seed=2)
        a = H.evolve(x0, t=2.5).to_numpy()
        b = H.evolve(x0, t=2.5, algo='chebyshev').to_numpy()
        assert np.max(np.abs(a - b)) < 1e-9, name
        H.destroy_mat()


def test_evolve_options_and_errors(golden_full):
    g = golden_full["mbl_L12"]
    H = models.mbl(12)
    x0 = State(L=12, state='random', seed=0)
    _evolve_check(H, x0, 1.0, g["expm_t=1"], ncv=10)
    _evolve_check(H, x0, 1.0, g["expm_t=1"], ncv=40, tol=1e-12)
    res = State(L=12)
    out = H.evolve(x0, t=1.0, result=res)
    assert out is res and res.initialized
    z = H.evolve(x0, t=0.0)
    assert np.array_equal(z.to_numpy(), x0.to_numpy())
    with pytest.raises(MaxIterationsError):       # test_evolve.py:196-202
        H.evolve(x0, t=500.0, max_its=2, ncv=5)
    with pytest.raises(ValueError):
        H.evolve(State(L=12, subspace=Parity('even'), state='random', seed=1), t=1.0)
    with pytest.raises(UninitializedError):
        H.evolve(State(L=12), t=1.0)
    # backwards in time undoes forwards
    back = H.evolve(H.evolve(x0, t=0.7), t=-0.7)
    assert np.max(np.abs(back.to_numpy() - x0.to_numpy())) < 1e-8


def test_evolve_pi_pulse():
    """test_evolve.py:23-32: exp(-i (pi/2) sigma_x) flips every spin."""
    L = 8
    H = models.xsum(L)
    y = H.evolve(State(L=L, state='U' * L), t=np.pi / 2)
    a = y.to_numpy()
    assert abs(abs(a[(1 << L) - 1]) - 1) < 1e-9


def test_evolve_subspace(golden_sub):
    L = 10
    H = models.mbl(L)
    sub = SpinConserve(L, 5)
    H.add_subspace(sub)
    x0 = State(L=L, subspace=sub, state='random', seed=0)
    y = H.evolve(x0, t=1.0).to_numpy()
    A = H.to_numpy(subspaces=(sub, sub)).toarray()
    w, U = np.linalg.eigh(A)
    ref = U @ (np.exp(-1j * w) * (U.conj().T @ x0.to_numpy()))
    assert abs(1 - np.vdot(ref, y) / np.vdot(ref, ref)) < 1e-9


# ------------------------------------------------------------------ eigsolve

def _check_eigs(H, evals, evecs, tol=1e-12, evec_tol=1e-11):
    """test_eigsolve.py:17-88."""
    for i, (ev, v) in enumerate(zip(evals, evecs)):
        Hv = H.dot(v)
        rq = Hv.dot(v).real          # sum Hv_i conj(v_i) = <v|Hv>
        assert abs(v.norm() - 1) < 1e-10
        assert abs(rq - ev) < max(tol, abs(ev) * tol) * 10
        r = Hv.copy()
        r.axpy(-ev, v)
        assert r.norm() < evec_tol * max(1.0, abs(ev)) * 10
        for j in range(i):
            assert abs(v.dot(evecs[j])) < 1e-9


@pytest.mark.parametrize("name,L", [("mbl", 12), ("mbl", 10), ("heisenberg", 10), ("long_range", 8),
                                    ("localized", 10), ("syk", 5), ("ising", 10)])
def test_eigsolve_golden(golden_full, name, L):
    g = golden_full[f"{name}_L{L}"]
    H = models.BY_NAME[name](L)
    ev, vecs = H.eigsolve(nev=5, which='lowest', tol=1e-12, getvecs=True)
    assert len(ev) >= 5
    # degenerate levels may be missed by Krylov solvers (computations.py:137-139):
    # every returned value must be an eigenvalue, and the lowest must match
    ref = g["evals_lowest"]
    assert abs(ev[0] - ref[0]) < 1e-9
    _check_eigs(H, ev[:5], vecs[:5])
    hi = H.eigsolve(nev=2, which='highest', tol=1e-10)
    assert abs(hi[0] - g["evals_highest"][0]) < 1e-8
    ex = H.eigsolve(nev=1, which='exterior', tol=1e-10)
    both = np.concatenate([g["evals_lowest"], g["evals_highest"]])
    assert abs(abs(ex[0]) - np.max(np.abs(both))) < 1e-8


def test_eigsolve_baseline_values(known):
    b = known["baseline_mbl_L12"]
    ev = models.mbl(12).eigsolve(nev=5, tol=1e-12)
    assert np.allclose(ev[:5], b["evals_lowest"], atol=5e-8)


def test_eigsolve_analytic_xsum():
    """test_eigsolve.py:95-123: lowest eigenvalue of sum sigma_x is -L, then -L+2."""
    L = 8
    ev = models.xsum(L).eigsolve(nev=2, tol=1e-12)
    assert abs(ev[0] + L) < 1e-10


def test_eigsolve_subspaces_and_errors():
    L = 12
    H = models.mbl(L)
    sub = SpinConserve(L, 6)
    H.add_subspace(sub)
    ev, vecs = H.eigsolve(nev=3, getvecs=True, tol=1e-12, subspace=sub)
    A = H.to_numpy(subspaces=(sub, sub)).toarray()
    w = np.linalg.eigvalsh(A)
    assert np.max(np.abs(ev[:3] - w[:3])) < 1e-9
    assert vecs[0].subspace == sub
    with pytest.raises(MaxIterationsError):      # test_eigsolve.py:231-241
        models.mbl(12).eigsolve(nev=8, max_its=1, tol=1e-14, ncv=9)
    with pytest.raises(RuntimeError):
        H.eigsolve(target=0.0)
    with pytest.raises(ValueError):
        H.eigsolve(which='target')
    with pytest.raises(ValueError):
        H.eigsolve(subspace=Parity('even', L=L))
    # zero-diagonal operator (test_eigsolve.py:158-163)
    Z = models.xsum(8)
    assert abs(Z.eigsolve(nev=1, which='highest', tol=1e-12)[0] - 8) < 1e-9


def test_operator_dot_api():
    L = 10
    H = models.ising(L)
    x = State(L=L, state='random', seed=3)
    y = H.dot(x)
    ref = H.to_numpy() @ x.to_numpy()
    assert np.max(np.abs(y.to_numpy() - ref)) < 1e-12
    y2 = H * x
    assert np.array_equal(y2.to_numpy(), y.to_numpy())
    assert abs(H.expectation(x) - np.vdot(x.to_numpy(), ref).real) < 1e-12
    assert abs(H.infinity_norm() - abs(H.to_numpy()).sum(axis=1).max()) < 1e-12
    # projection gate (operators.py:598-603)
    H.add_subspace(SpinConserve(L, 5))
    with pytest.raises(ValueError):
        H.build_mat()
    H.allow_projection = True
    H.build_mat()
    with pytest.raises(ValueError):
        Operator(msc=[(1, 0, 1j)]).dot(State(L=1, state='U'))   # non-Hermitian


def test_conserves_gpu():
    """CheckConserves on the device against the oracle's restatement
    (tests/integration/test_operators.py:20-188 cases in spirit)."""
    from oracle import oracle as orc
    from gpu_util import orc_msc, orc_sub
    cases = [(models.heisenberg(10), SpinConserve(10, 5), True), (models.heisenberg(10), Parity('even', L=10), True),
             (models.xsum(8), SpinConserve(8, 4), False), (models.ising(8), SpinConserve(8, 4), False),
             (models.ising(8), Parity('odd', L=8), False), (models.xxz(8), Parity('odd', L=8), True),
             (models.long_range(8), Parity('even', L=8), False), (models.mbl(12), SpinConserve(12, 3), True)]
    for H, sub, want in cases:
        got = H.conserves(sub)
        assert got == want
        assert got == orc.check_conserves(orc_msc(H), orc_sub(sub), orc_sub(sub))
    # projection between different subspaces
    H = models.heisenberg(8)
    assert H.conserves(Full(L=8), SpinConserve(8, 4))          # sector into full space: always inside
    assert not H.conserves(SpinConserve(8, 4), Full(L=8))      # full space into one sector: leaves it
    assert not H.conserves(SpinConserve(8, 3), SpinConserve(8, 4))


def _state_from(sub, arr):
    st = State(L=sub.L, subspace=sub)
    st.vec.set_local_from_numpy(np.ascontiguousarray(arr, dtype=complex))
    st.set_initialized()
    return st


def _xp_parent(name, L):
    if "_sc" in name:
        return SpinConserve(L, L // 2)
    if "parity" in name:
        return Parity("even" if "even" in name else "odd", L=L)
    return Full(L=L)


def test_xparity_golden_gpu(golden_xp, monkeypatch):
    """Operator.dot / infinity_norm / conserves / eigsolve on XParity subspaces against the
    fixtures made with the reference's XParity.reduce_msc + msc_to_numpy
    (tests/integration/test_multiply.py XParity cases in spirit)."""
    monkeypatch.setenv("DNM_TILE_BITS", "8")
    monkeypatch.setenv("DNM_LOG_ROWS", "2")
    for name in golden_xp.names():
        g = golden_xp[name]
        L, sector = int(g["L"]), int(g["sector"])
        hname = next(k for k in models.BY_NAME if name.startswith(k))
        H = models.BY_NAME[hname](L)
        sub = XParity(_xp_parent(name, L), sector=sector)
        H.add_subspace(sub)
        conserved = bool(g["conserved"]) and not ("ising" in name and "parity" in name)
        assert H.conserves(sub) == conserved, name
        if not conserved:
            with pytest.raises(ValueError):
                H.build_mat()
            H.allow_projection = True
        y = H.dot(_state_from(sub, g["x"]))
        assert y.subspace is sub
        nnz = g["coeffs"].size
        assert np.max(np.abs(y.to_numpy() - g["y"])) <= 8 * nnz * 2.2e-16 * max(1.0, np.abs(g["coeffs"]).max()), name
        nrm = H.infinity_norm()
        assert abs(nrm - float(g["infnorm"])) <= nnz * 2.2e-16 * 100 * max(1.0, nrm), name
        assert np.max(np.abs(H.to_numpy().toarray() @ g["x"] - g["y"])) < 1e-12
        if conserved:
            ev = H.eigsolve(nev=1, tol=1e-11)
            assert abs(ev[0] - g["evals_lowest"][0]) < 1e-9, name
        H.destroy_mat()


@pytest.mark.parametrize("sub", ["full", "sc"])
def test_eigsolve_beta_branches(monkeypatch, sub):
    """The three ways a three-term step gets its beta (fused: |p|^2 - |alpha|^2 before the update; norm sweep
    after it; fused plus the corrective rescaling) must give the same spectrum and residuals."""
    L = 14
    s = Full(L=L) if sub == "full" else SpinConserve(L, L // 2)
    got = []
    for mode in ("", "sweep", "rescale"):
        if mode:
            monkeypatch.setenv("DNM_EIGS_BETA", mode)
        if sub == "sc":
            monkeypatch.setenv("DNM_SC_BLOCK", "10")
        H = models.mbl(L)
        H.add_subspace(s)
        evals, evecs = H.eigsolve(nev=3, getvecs=True, tol=1e-11, subspace=s)
        for e, v in zip(evals[:3], evecs[:3]):
            r = H.dot(v)
            r.axpy(-e, v)
            assert r.norm() < 1e-9
        got.append(np.array(evals[:3]))
        H.destroy_mat()
    assert np.max(np.abs(got[0] - got[1])) < 1e-10 and np.max(np.abs(got[0] - got[2])) < 1e-10


@pytest.mark.parametrize("sector", [+1, -1])
def test_xparity_spinconserve_block_kernel(monkeypatch, sector):
    """XParity(SpinConserve): the reduced operator (complemented many-spin masks next to the chain bonds) through
    the block form of the SpinConserve kernel, the row kernel and the host-built sparse matrix."""
    L = 16
    sub = XParity(SpinConserve(L, L // 2), sector=sector)
    outs = []
    for blk in ("10", "0"):
        monkeypatch.setenv("DNM_SC_BLOCK", blk)
        H = models.heisenberg(L)
        H.add_subspace(sub)
        x = State(subspace=sub, state='random', seed=4)
        assert ("block form" in H.get_mat().describe()) == (blk == "10")
        outs.append(H.dot(x).to_numpy())
        if blk == "10":
            want = H.to_numpy().toarray() @ x.to_numpy()
        H.destroy_mat()
    assert np.max(np.abs(outs[0] - outs[1])) < 1e-13
    assert np.max(np.abs(outs[0] - want)) < 1e-12


@pytest.mark.parametrize("parent", ["full", "sc", "parity"])
@pytest.mark.parametrize("sector", [+1, -1])
def test_xparity_convert_state(parent, sector):
    """convert_state (subspaces.py:676-762) is the isometry between the sector and its
    parent: norm-preserving, inverted by the way back, and it intertwines the reduced and
    the parent operator for a Hamiltonian that commutes with the global flip."""
    L = 10
    par = {"full": Full(L=L), "sc": SpinConserve(L, 5), "parity": Parity("even", L=L)}[parent]
    sub = XParity(par, sector=sector)
    H = models.heisenberg(L)
    H.add_subspace(sub)
    H.add_subspace(par)
    psi = State(L=L, subspace=sub, state="random", seed=4)
    up = sub.convert_state(psi)
    assert up.subspace is par and abs(up.norm() - 1) < 1e-13
    v = up.to_numpy()
    flipped = par.state_to_idx(par.idx_to_state(np.arange(par.get_dimension())) ^ ((1 << L) - 1))
    assert np.max(np.abs(v[flipped] - sector * v)) < 1e-15          # eigenvector of the global flip
    back = sub.convert_state(up)
    assert np.max(np.abs(back.to_numpy() - psi.to_numpy())) < 1e-14
    lhs = sub.convert_state(H.dot(psi)).to_numpy()
    rhs = H.dot(up).to_numpy()
    assert np.max(np.abs(lhs - rhs)) < 1e-12
    with pytest.raises(ValueError):
        sub.convert_state(State(L=L, subspace=Full(L=L) if parent != "full" else Parity("odd", L=L),
                                state="random", seed=1))


# ------------------------------------------------------------------ reduced density matrix / entropies

def test_rdm_known_answers_gpu(known):
    """tests/integration/test_rdm.py: error cases (:70-86), empty keep (:88-103), complex
    sign (:105-122), the L=4 tables with entanglement / Renyi entropies (:124-192)."""
    from conftest import cplx, cmatrix
    from dynamite_amd.computations import reduced_density_matrix, renyi_entropy
    r = known["rdm"]
    st = State(L=4, state='U' * 4)
    for bad in ([-1, 0], [0, 1, 50], [1, 0]):
        with pytest.raises(ValueError):
            reduced_density_matrix(st, bad)
    e = r["empty_keep"]
    s2 = _state_from(Full(L=2), [cplx(v) for v in e["state"]])
    assert np.array_equal(reduced_density_matrix(s2, []), cmatrix(e["dm"]))
    cs = r["complex_sign"]
    s2 = _state_from(Full(L=2), [cplx(v) for v in cs["state"]])
    assert np.allclose(reduced_density_matrix(s2, cs["keep"]), cmatrix(cs["dm"]), atol=1e-15)
    s4 = _state_from(Full(L=4), [cplx(v) for v in r["L4"]["state"]])
    for c in r["L4"]["cases"]:
        dm = reduced_density_matrix(s4, c["keep"])
        assert np.allclose(dm, cmatrix(c["dm"]), atol=2e-6, rtol=0)
        assert abs(s4.entanglement_entropy(c["keep"]) - c["entropy"]) < 1e-5
        assert abs(renyi_entropy(s4, c["keep"], 1) - c["entropy"]) < 1e-5
        assert abs(renyi_entropy(s4, c["keep"], 0) - np.log(2 ** len(c["keep"]))) < 1e-12
    # product state: zero entropy, projector RDM (test_rdm.py:209-216)
    sp = State(L=6, state=0b010011)
    dm = reduced_density_matrix(sp, [0, 1])
    want = np.zeros((4, 4)); want[3, 3] = 1
    assert np.array_equal(dm, want) and sp.entanglement_entropy([0, 1, 2]) == 0
    with pytest.raises(ValueError):          # test_rdm.py:289-303
        reduced_density_matrix(State(L=6, subspace=XParity(SpinConserve(6, 3)), state='random', seed=0), [0])


@pytest.mark.parametrize("kind", ["full", "even", "odd", "sc", "explicit"])
def test_rdm_vs_oracle_and_reshape(kind):
    """Every subspace type, every tile size of the kernel (k = 1..9), contiguous and scattered
    keep sets: against the oracle restatement of rdm_<SUBSPACE> and against the reference
    tests' own check (embed in the full space, reshape, multiply: test_rdm.py:218-270)."""
    from oracle import oracle as orc
    from gpu_util import orc_sub
    from dynamite_amd.computations import reduced_density_matrix
    L = 12
    rs = np.random.RandomState(5)
    sub = {"full": Full(L=L), "even": Parity('even', L=L), "odd": Parity('odd', L=L), "sc": SpinConserve(L, 6),
           "explicit": Explicit(np.sort(rs.choice(1 << L, 900, replace=False)), L=L)}[kind]
    st = State(L=L, subspace=sub, state='random', seed=2)
    v = st.to_numpy()
    full = np.zeros(1 << L, dtype=complex)
    full[sub.idx_to_state(np.arange(sub.get_dimension()))] = v
    osub = orc_sub(sub)
    for n in range(1, L - 2):
        m = full.reshape((full.size // 2 ** n, -1))
        dm = reduced_density_matrix(st, list(range(n, L))) if L - n <= 9 else None
        if dm is not None:
            assert np.max(np.abs(dm - m @ m.conj().T)) < 1e-14
        if n <= 9:
            dm = reduced_density_matrix(st, list(range(n)))
            assert np.max(np.abs(dm - m.T @ m.conj())) < 1e-14
    for keep in ([0], [L - 1], [1, 4, 6], [0, 2, 3, 7, 8, 11], [2, 3, 4, 5, 6, 7, 8, 9]):
        dm = reduced_density_matrix(st, keep)
        assert np.max(np.abs(dm - orc.rdm(osub, v, keep))) < 1e-14
        assert abs(np.trace(dm) - 1) < 1e-13 and np.allclose(dm, dm.conj().T, atol=1e-17, rtol=0)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_FUZZ_RDM_N", "24"))))
def test_fuzz_rdm(seed):
    """Random sizes, subspaces and SCATTERED keep sets of 1..10 spins (the streaming kernel for one / two spins, the
    vector-unit tiles, the MFMA tiles from six spins on with several tiles and slices) against the oracle's restatement
    of rdm_<SUBSPACE> (bpetsc_template_1.c:87-165)."""
    from oracle import oracle as orc
    from gpu_util import orc_sub
    from dynamite_amd.computations import reduced_density_matrix
    rs = np.random.RandomState(9000 + seed)
    L = int(rs.randint(6, 17))
    kind = ["full", "even", "odd", "sc", "explicit"][rs.randint(5)]
    sub = {"full": lambda: Full(L=L), "even": lambda: Parity('even', L=L), "odd": lambda: Parity('odd', L=L),
           "sc": lambda: SpinConserve(L, int(rs.randint(1, L))),
           "explicit": lambda: Explicit(np.sort(rs.choice(1 << L, min(1 << L, int(rs.randint(2, 3000))), replace=False)), L=L)}[kind]()
    st = State(L=L, subspace=sub, state='random', seed=seed)
    v = st.to_numpy()
    osub = orc_sub(sub)
    for _ in range(4):
        k = int(rs.randint(1, min(L, 10) + 1))
        keep = sorted(int(i) for i in rs.choice(L, size=k, replace=False))
        dm = reduced_density_matrix(st, keep)
        want = orc.rdm(osub, v, np.array(keep, dtype=np.int64))
        assert dm.shape == want.shape and np.max(np.abs(dm - want)) < 1e-14, (L, kind, keep)
        assert np.allclose(dm, dm.conj().T, atol=1e-17, rtol=0)


def test_rdm_large_properties():
    """L = 24 (2^24 amplitudes): trace, Hermiticity, S(A) = S(complement) for a pure state,
    and agreement with the host on a sampled 2-spin block."""
    from dynamite_amd.computations import reduced_density_matrix, dm_entanglement_entropy
    L = 24
    st = State(L=L, state='random', seed=9)
    a = reduced_density_matrix(st, list(range(8)))
    assert abs(np.trace(a) - 1) < 1e-12 and np.allclose(a, a.conj().T, atol=1e-17, rtol=0)
    v = st.to_numpy()
    m = v.reshape((1 << 16, 1 << 8))
    assert np.max(np.abs(a - m.T @ m.conj())) < 1e-13
    b = reduced_density_matrix(st, [3, 20])
    # axes of the reshape: spins 23..21 | 20 | 19..4 | 3 | 2..0; rows of the RDM = (spin 20, spin 3)
    m2 = v.reshape(8, 2, 1 << 16, 2, 8).transpose(1, 3, 0, 2, 4).reshape(4, -1)
    assert np.max(np.abs(b - m2 @ m2.conj().T)) < 1e-13
    st = State(L=20, state='random', seed=10)       # (the host eigensolver bounds the size of this one)
    sA = dm_entanglement_entropy(reduced_density_matrix(st, list(range(9))))
    sB = dm_entanglement_entropy(reduced_density_matrix(st, list(range(9, 20))))
    assert abs(sA - sB) < 1e-9 and 0 < sA <= 9 * np.log(2)


def test_entropies_device_spectrum():
    """entanglement_entropy / renyi_entropy diagonalise reduced density matrices of 256 x 256 and more on the device
    (torch.linalg.eigvalsh on the tensor the RDM kernel wrote -- no copy of the matrix to the host): the same numbers as
    the reference's route, numpy on the host array (computations.py:351-454), for every alpha branch."""
    from dynamite_amd import computations as cp
    assert cp._DEVICE_EIG_FROM == 256
    st = State(L=20, state='random', seed=21)
    for keep in (list(range(8)), [1, 2, 3, 5, 8, 11, 12, 15, 17], list(range(10, 20))):
        dm = cp.reduced_density_matrix(st, keep)
        assert dm.shape == (1 << len(keep),) * 2
        want = cp.dm_entanglement_entropy(dm)
        assert abs(cp.entanglement_entropy(st, keep) - want) < 1e-11
        for alpha in (0, 1, 2, 2.5, 'inf'):
            assert abs(cp.renyi_entropy(st, keep, alpha) - cp.dm_renyi_entropy(dm, alpha)) < 1e-10, alpha
        assert abs(cp.renyi_entropy(st, keep, 3, method='matrix_power') - cp.dm_renyi_entropy(dm, 3, 'eigsolve')) < 1e-10
    # a state of fixed magnetisation: the spectrum is taken block by block (equal numbers of up spins among the kept)
    for sub_, mask in ((SpinConserve(18, 9), ~0), (Parity('odd', L=16), 1)):
        ss = State(L=sub_.L, subspace=sub_, state='random', seed=22)
        for keep in (list(range(9)), [0, 2, 3, 5, 8, 11, 12, 13, 15], list(range(sub_.L - 10, sub_.L))):
            dm = cp.reduced_density_matrix(ss, keep)
            ones = np.array([bin(i).count("1") for i in range(dm.shape[0])]) & mask
            assert np.abs(dm[ones[:, None] != ones[None, :]]).max() == 0.0           # (what the block form relies on)
            assert abs(cp.entanglement_entropy(ss, keep) - cp.dm_entanglement_entropy(dm)) < 1e-11
            for alpha in (0, 2, 'inf'):
                assert abs(cp.renyi_entropy(ss, keep, alpha) - cp.dm_renyi_entropy(dm, alpha)) < 1e-10, alpha
    with pytest.raises(ValueError):
        cp.entanglement_entropy(st, list(range(9, 0, -1)))        # the reference's argument checks hold on this route too
    with pytest.raises(ValueError):
        cp.renyi_entropy(st, list(range(12, 21)), 2)


# ------------------------------------------------------------------ files

def test_state_and_operator_files(tmp_path):
    """State.save / from_file (states.py:627-701: <name>.metadata + PETSc binary <name>.vec) and
    Operator.save / load (operators.py:505-542)."""
    import pickle
    import struct
    for sub in (Full(L=9), SpinConserve(10, 4), Parity('odd', L=8), XParity(SpinConserve(8, 4), '-'),
                Explicit([5, 3, 12, 9], L=4)):
        st = State(L=sub.L, subspace=sub, state='random', seed=1)
        for int_size in (64, 32):
            fn = str(tmp_path / ("s%d" % int_size))
            st.save(fn, int_size=int_size)
            raw = open(fn + '.vec', 'rb').read()
            hb = int_size // 8
            cid, n = struct.unpack('>qq' if int_size == 64 else '>ii', raw[:2 * hb])
            assert (cid, n) == (1211214, sub.get_dimension()) and len(raw) == 2 * hb + 16 * n
            assert np.array_equal(np.frombuffer(raw[2 * hb:], dtype='>c16'), st.to_numpy())
            back = State.from_file(fn)
            assert back.subspace == sub and type(back.subspace) is type(sub)
            assert np.array_equal(back.to_numpy(), st.to_numpy())
    # metadata as dynamite writes it: a pickled dynamite.subspaces object (attribute names of subspaces.py)
    import sys, types
    fake = types.ModuleType('dynamite.subspaces')
    for name in ('Parity', 'XParity', 'SpinConserve'):
        setattr(fake, name, type(name, (), {'__module__': 'dynamite.subspaces'}))
    sys.modules.setdefault('dynamite', types.ModuleType('dynamite'))
    sys.modules['dynamite.subspaces'] = fake
    try:
        par = fake.SpinConserve(); par.__dict__.update(_L=8, _k=4, _chksum=None)
        xp = fake.XParity(); xp.__dict__.update(_parent=par, _sector=-1)
        with open(fn + '.metadata', 'wb') as f:
            pickle.dump(xp, f)
    finally:
        del sys.modules['dynamite.subspaces']
    sub = XParity(SpinConserve(8, 4), '-')
    st = State(L=8, subspace=sub, state='random', seed=2)
    st.save(str(tmp_path / "x"))
    import shutil
    shutil.copy(fn + '.metadata', str(tmp_path / "x.metadata"))
    back = State.from_file(str(tmp_path / "x"))
    assert back.subspace == sub and np.array_equal(back.to_numpy(), st.to_numpy())
    # corrupt: wrong length for the subspace
    Full(L=3)
    with open(str(tmp_path / "x.metadata"), 'wb') as f:
        pickle.dump(Full(L=3), f)
    with pytest.raises(RuntimeError):
        State.from_file(str(tmp_path / "x"))
    H = models.long_range(7)
    H.save(str(tmp_path / "op"))
    H2 = Operator.load(str(tmp_path / "op"))
    assert np.array_equal(H2.msc, H.msc) and H2 == H


def test_state_project_and_function():
    """State.project (states.py:364-402; tests/integration/test_states.py projection cases) and
    set_all_by_function (:320-360)."""
    for sub in (Full(L=10), SpinConserve(10, 5), Parity('even', L=10)):
        st = State(L=10, subspace=sub, state='random', seed=3)
        v = st.to_numpy()
        sts = sub.idx_to_state(np.arange(sub.get_dimension()))
        for index, value in ((0, 1), (7, 0)):
            p = st.copy()
            p.project(index, value)
            ref = np.where(((sts >> index) & 1) == value, v, 0)
            ref = ref / np.linalg.norm(ref)
            assert np.max(np.abs(p.to_numpy() - ref)) < 1e-15
        with pytest.raises(ValueError):
            st.project(10, 0)
        with pytest.raises(ValueError):
            st.project(0, 2)
        f = State(L=10, subspace=sub)
        f.set_all_by_function(lambda s: (s % 7) + 1j * (s & 3), vectorize=True)
        g = State(L=10, subspace=sub)
        g.set_all_by_function(lambda s: (s % 7) + 1j * (s & 3))
        assert np.array_equal(f.to_numpy(), (sts % 7) + 1j * (sts & 3)) and np.array_equal(f.to_numpy(), g.to_numpy())
    a = State(L=6, state='random', seed=1)
    b = 2 - a
    assert np.allclose(b.to_numpy(), 2 - a.to_numpy(), atol=1e-15)


# ------------------------------------------------------------------ BASELINE config 2 size

def test_krylov_full_size_properties():
    """L=26 XXZ, 2^26 amplitudes (BASELINE configs[1]): evolve conserves the norm and the energy and is
    undone by the reverse evolution; the eigsolve result satisfies the residual bound it promises."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 16 * (1 << 26):
        pytest.skip("not enough HBM")
    L = 26
    H = models.xxz(L)
    x = State(L=L, state='random', seed=5)
    e0 = H.expectation(x)
    y = H.evolve(x, t=0.7)
    assert abs(y.norm() - 1) < 1e-9
    assert abs(H.expectation(y) - e0) < 1e-7
    z = H.evolve(y, t=-0.7)
    z.axpy(-1.0, x)
    assert z.norm() < 1e-7
    yi = H.evolve(x, t=-0.2j)                    # imaginary time lowers the energy
    yi.normalize()
    assert H.expectation(yi) < e0
    ev, vecs = H.eigsolve(nev=1, getvecs=True, tol=1e-9)
    v = vecs[0]
    r = H.dot(v)
    r.axpy(-ev[0], v)
    assert abs(v.norm() - 1) < 1e-10 and r.norm() < 1e-8 * abs(ev[0])
    from dynamite_amd.computations import eigsolve as _es
    assert abs(_es.last_stats['max_rel_residual'] - r.norm() / abs(ev[0])) < 1e-12      # the solver measured the same
    assert ev[0] < H.expectation(yi) < e0


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_FUZZ_KRYLOV_N", "6"))))
def test_fuzz_krylov_random_operators(seed):
    """evolve (real, imaginary and complex times) and eigsolve on random Pauli-string Hamiltonians and
    subspaces against dense linear algebra on the host."""
    from test_gpu_matvec import _random_hermitian
    rs = np.random.RandomState(500 + seed)
    L = int(rs.randint(6, 11))
    H = _random_hermitian(L, int(rs.randint(4, 25)), rs)
    kind = ["full", "parity", "xparity"][rs.randint(3)]
    sub = Full(L=L)
    if kind == "parity":
        sub = Parity(int(rs.randint(2)), L=L)
    elif kind == "xparity":
        sub = XParity(Full(L=L), sector=[+1, -1][rs.randint(2)])
    H.add_subspace(sub)
    H.allow_projection = True
    A = H.to_numpy(subspaces=(sub, sub)).toarray()
    A = (A + A.conj().T) / 2 if kind != "full" else A      # projections of a Hermitian operator stay Hermitian
    w, U = np.linalg.eigh(A)
    x = State(L=L, subspace=sub, state='random', seed=seed)
    for t in (float(rs.uniform(0.1, 2.0)), -1j * float(rs.uniform(0.05, 0.5)), complex(rs.uniform(0.1, 1), -rs.uniform(0.05, 0.3))):
        y = H.evolve(x, t=t, tol=1e-10).to_numpy()
        ref = U @ (np.exp(-1j * t * w) * (U.conj().T @ x.to_numpy()))
        assert np.linalg.norm(y - ref) < 1e-8 * max(1.0, np.linalg.norm(ref)), (kind, t)
    if np.ptp(w) > 1e-6:
        ev = H.eigsolve(nev=2, tol=1e-11, subspace=sub)
        assert abs(ev[0] - w[0]) < 1e-9
        hi = H.eigsolve(nev=1, which='highest', tol=1e-11, subspace=sub)
        assert abs(hi[0] - w[-1]) < 1e-9


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_FUZZ_EIGS_REAL_N", "12"))))
def test_fuzz_eigsolve_real_arithmetic(monkeypatch, seed):
    """eigsolve in real arithmetic (forced) on random real symmetric Pauli sums, Full and Parity, through the restarted,
    the basis-free and the filtered driver, at both ends: eigenvalues against dense diagonalisation, the returned complex
    states against the reference's residual bar (tests/integration/test_eigsolve.py:17-88)."""
    from test_gpu_matvec import _random_real_symmetric
    from dynamite_amd.computations import eigsolve
    rs = np.random.RandomState(8000 + seed)
    L = int(rs.randint(10, 13))              # (the dense reference on the host sets the size)
    for k, v in (("DNM_TILE_BITS", "8"), ("DNM_LOG_ROWS", "2"), ("DNM_PLAN_MODE", str(int(rs.randint(3)))), ("DNM_GBITS", "3"),
                 ("DNM_AMIN", "3"), ("DNM_EIGS_REAL", "1")):
        monkeypatch.setenv(k, v)
    mode = ["restarted", "basis_free", "filtered"][rs.randint(3)]
    if mode == "basis_free":
        monkeypatch.setenv("DNM_EIGS_BASISFREE", "1")
    elif mode == "filtered":
        monkeypatch.setenv("DNM_EIGS_FILTER", "1")
    H = _random_real_symmetric(L, int(rs.randint(6, 30)), rs)
    sub = Full(L=L) if rs.randint(2) else Parity(int(rs.randint(2)), L=L)
    H.add_subspace(sub)
    H.allow_projection = True
    A = H.to_numpy(subspaces=(sub, sub)).toarray()
    A = (A + A.conj().T) / 2
    if np.abs(A.imag).max() > 0:
        return                                   # (the projection onto a parity sector of a real operator stays real; guard)
    w = np.linalg.eigvalsh(A)
    if np.ptp(w) < 1e-6:
        return
    which = ["lowest", "highest"][rs.randint(2)]
    nev = int(rs.randint(1, 4))              # (basis-free with nev > 1: one pair after the other by deflation)
    ev, vecs = H.eigsolve(nev=nev, which=which, tol=1e-10, subspace=sub, getvecs=True)
    if eigsolve.last_stats['real_arithmetic'] is not True:
        # a parity projection that drops terms can leave an operator the packed form refuses; then complex ran
        assert isinstance(sub, Parity)
    want = w[0] if which == "lowest" else w[-1]
    assert abs(ev[0] - want) < 1e-8 * max(1.0, abs(want)), (L, mode, which)
    for e, vs in zip(ev, vecs):
        v = vs.to_numpy()
        assert np.min(np.abs(w - e)) < 1e-8 * max(1.0, abs(e))                    # every returned value is an eigenvalue
        assert np.linalg.norm(A @ v - e * v) < 1e-7 * max(1.0, abs(e)), (L, mode, which)
        assert abs(np.linalg.norm(v) - 1) < 1e-10
    H.destroy_mat()


def test_tools_memory():
    from dynamite_amd import tools
    tools.track_memory()
    before = tools.get_memory_usage(group_by='rank')
    st = State(L=24, state='random', seed=0)           # 256 MiB of HBM
    after = tools.get_memory_usage(group_by='rank')
    assert after - before > 0.2 and tools.get_memory_usage(max_usage=True) > 0.2
    assert 'dynamite_amd' in tools.get_version() and 'ABI' in tools.get_version_str()
    with pytest.raises(ValueError):
        tools.get_memory_usage(group_by='socket')
    del st


def test_floquet_script_flow(tmp_path):
    """The flow of the reference's examples/scripts/floquet/run_floquet.py on this engine (global
    config.L, operator products, evolve into a result state, pi pulse by an operator, expectation values
    with a scratch state, half-chain entropy with a range, checkpoint save / resume), checked against dense
    linear algebra."""
    from dynamite_amd import config
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, index_sum, index_product, op_sum
    from dynamite_amd.computations import entanglement_entropy, dm_entanglement_entropy
    from dynamite_amd.tools import MPI_COMM_WORLD
    old = config.L
    try:
        L = config.L = 8
        alpha, Jx, h, T = 1.25, 0.19, [0.21, 0.17, 0.13], 0.12
        H = (op_sum(1 / r ** alpha * index_sum(0.25 * sigmaz(0) * sigmaz(r)) for r in range(1, L))
             + Jx * index_sum(0.25 * sigmax(0) * sigmax(1))
             + index_sum(op_sum(hi * 0.5 * s() for hi, s in zip(h, [sigmax, sigmay, sigmaz]))))
        X = index_product(sigmax())
        Deff = (H + X * H * X) / 2
        Sz = [0.5 * sigmaz(i) for i in range(L)]
        state = State(state='U' * 4 + 'D' * 4)
        tmp = state.copy()
        Hd, Xd, Dd = (O.to_numpy().toarray() for O in (H, X, Deff))
        assert np.allclose(Dd, (Hd + Xd @ Hd @ Xd) / 2, atol=1e-14)
        w, U = np.linalg.eigh(Hd)
        step = Xd @ (U @ np.diag(np.exp(-1j * T * w)) @ U.conj().T)
        ref = state.to_numpy()
        for cycle in range(1, 7):
            H.evolve(state, result=tmp, t=T)
            X.dot(tmp, result=state)
            ref = step @ ref
            assert np.linalg.norm(state.to_numpy() - ref) < 1e-8
            e = Deff.expectation(state, tmp_state=tmp)
            assert abs(e - np.vdot(ref, Dd @ ref).real) < 1e-8
            ent = entanglement_entropy(state, keep=range(L // 2))
            m = ref.reshape(1 << (L - L // 2), 1 << (L // 2))
            assert abs(ent - dm_entanglement_entropy(m.T @ m.conj())) < 1e-8
            sz = [Sz[i].expectation(state, tmp_state=tmp) for i in range(L)]
            bits = (np.arange(1 << L)[:, None] >> np.arange(L)) & 1
            assert np.allclose(sz, (np.abs(ref) ** 2) @ (0.5 - bits), atol=1e-8)
            if cycle == 3:
                state.save(str(tmp_path / "floquet_cycle_3"))
        assert MPI_COMM_WORLD().rank == 0 and MPI_COMM_WORLD().size == 1
        resumed = State.from_file(str(tmp_path / "floquet_cycle_3"))
        for cycle in range(4, 7):
            H.evolve(resumed, result=tmp, t=T)
            X.dot(tmp, result=resumed)
        assert np.linalg.norm(resumed.to_numpy() - state.to_numpy()) < 1e-12
    finally:
        config.L = old


def test_mbl_script_flow():
    """The flow of the reference's examples/scripts/MBL/run_mbl.py (global config.L and config.subspace,
    Python sum() of operators, extremal eigenpairs with vectors, half-chain entropies and gap ratios) --
    without the interior `target` solves, which the reference refuses for shell matrices too."""
    from dynamite_amd import config
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, index_sum
    from dynamite_amd.computations import dm_entanglement_entropy
    oldL, olds = config.L, config.subspace
    try:
        L = config.L = 10
        sub = config.subspace = SpinConserve(L, L // 2)
        np.random.seed(0xB0BA)
        H = index_sum(0.25 * sum(s(0) * s(1) for s in [sigmax, sigmay, sigmaz]))
        H = H + sum(0.5 * np.random.uniform(-2.0, 2.0) * sigmaz(i) for i in range(L))
        A = H.to_numpy().toarray()
        assert A.shape == (252, 252)
        w, U = np.linalg.eigh(A)
        states = sub.idx_to_state(np.arange(252))
        for which, wref, cols in (('lowest', w[:4], U[:, :4]), ('highest', w[::-1][:4], U[:, ::-1][:, :4])):
            evals, evecs = H.eigsolve(nev=4, which=which, getvecs=True, tol=1e-11)
            assert np.allclose(evals[:4], wref, atol=1e-9)
            for i, v in enumerate(evecs[:4]):
                assert v.subspace == sub
                full = np.zeros(1 << L, dtype=complex)
                full[states] = cols[:, i]
                m = full.reshape(1 << (L - L // 2), 1 << (L // 2))
                assert abs(v.entanglement_entropy(keep=range(L // 2)) - dm_entanglement_entropy(m.T @ m.conj())) < 1e-7
            ev = sorted(evals[:4])
            ratio = np.mean([min(ev[i] - ev[i - 1], ev[i + 1] - ev[i]) / max(ev[i] - ev[i - 1], ev[i + 1] - ev[i])
                             for i in range(1, 3)])
            assert 0 < ratio <= 1
        with pytest.raises(RuntimeError):
            H.eigsolve(nev=2, target=0.0)
    finally:
        config.L, config.subspace = oldL, olds


def test_kagome_script_flow():
    """The flow of the reference's examples/scripts/kagome/run_kagome.py: Heisenberg couplings on an
    arbitrary graph (non-adjacent bonds), SpinConserve at half filling wrapped in XParity, `H.shell =
    False` as the script passes without --shell (accepted, still matrix-free), eigsolve(nev=2) -> gap."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum
    # a 12-site ring of corner-sharing triangles (kagome strip): triangles (2i, 2i+1, 2i+2)
    N = 12
    edges = set()
    for t in range(0, N, 2):
        a, b, c = t, (t + 1) % N, (t + 2) % N
        edges |= {(min(a, b), max(a, b)), (min(b, c), max(b, c)), (min(a, c), max(a, c))}
    H = op_sum(op_sum(0.25 * s(i) * s(j) for s in [sigmax, sigmay, sigmaz]) for i, j in sorted(edges))
    assert H.get_length() == N
    sub = XParity(SpinConserve(N, N // 2), sector=+1 if N % 4 == 0 else -1)
    H.subspace = sub
    with pytest.warns(UserWarning):
        H.shell = False
    assert H.shell is True
    gs, e1 = H.eigsolve(nev=2, tol=1e-11)[:2]
    w = np.linalg.eigvalsh(H.to_numpy().toarray())
    # (the lowest level of this cluster is degenerate; the second value is then its other copy or the next level)
    assert abs(gs - w[0]) < 1e-9 and e1 >= gs - 1e-9 and np.min(np.abs(w - e1)) < 1e-8
    # the sector's lowest level is a level of the parent's spectrum
    par = SpinConserve(N, N // 2)
    H.add_subspace(par)
    wp = np.linalg.eigvalsh(H.to_numpy(subspaces=(par, par)).toarray())
    assert np.min(np.abs(wp - gs)) < 1e-9


def test_syk_script_flow():
    """The flow of the reference's examples/scripts/SYK/run_syk.py: Majorana operators that map between the
    two Parity sectors (left != right subspaces), an SYK Hamiltonian carrying both sectors, imaginary-time
    cooling with copy(result=), and the out-of-time-order correlator by chained evolve / dot calls --
    against dense linear algebra in the full space."""
    from itertools import combinations
    from dynamite_amd import config
    from dynamite_amd.operators import op_sum, op_product
    from dynamite_amd.extras import majorana
    old = config.L
    try:
        N = 12
        L = config.L = (N + 1) // 2
        np.random.seed(7)
        even, odd = Parity('even'), Parity('odd')
        W, V = majorana(0), majorana(1)
        for O in (W, V):
            O.add_subspace(even, odd)
            O.add_subspace(odd, even)
        maj = [majorana(i) for i in range(N)]

        def products():
            for idxs in combinations(range(N), 4):
                p = op_product(maj[i] for i in idxs)
                p.scale(np.random.normal())
                yield p
        H = op_sum(products())
        H.scale(np.sqrt(6 / N ** 3))
        H.add_subspace(even)
        H.add_subspace(odd)

        full = Full(L=L)
        Hd, Wd, Vd = (O.to_numpy(subspaces=(full, full)).toarray() for O in (H, W, V))
        w, U = np.linalg.eigh(Hd)
        expH = lambda z: U @ np.diag(np.exp(z * w)) @ U.conj().T       # noqa: E731

        def embed(st):
            out = np.zeros(1 << L, dtype=complex)
            out[st.subspace.idx_to_state(np.arange(len(st)))] = st.to_numpy()
            return out

        psi0 = State(state='random', subspace=even, seed=3)
        psi1 = psi0.copy()
        ref0 = embed(psi0)
        H.evolve(psi0, t=-1j * 0.4, result=psi1)          # cool: exp(-0.4 H)
        psi1.normalize()
        psi1.copy(result=psi0)
        ref0 = expH(-0.4) @ ref0
        ref0 /= np.linalg.norm(ref0)
        assert np.linalg.norm(embed(psi0) - ref0) < 1e-8
        t = 0.9
        tmp_odd_0 = V * psi0
        assert tmp_odd_0.subspace == odd
        tmp_odd_1 = H.evolve(tmp_odd_0, t=t)
        W.dot(tmp_odd_1, result=psi0)
        tmp_even = H.evolve(psi0, t=-t)
        V.dot(tmp_even, result=tmp_odd_0)
        H.evolve(tmp_odd_0, t=t, result=tmp_odd_1)
        W.dot(tmp_odd_1, result=psi0)
        H.evolve(psi0, t=-t, result=tmp_even)
        got = 2 * psi1.dot(tmp_even).real + 0.5
        Wt = expH(1j * t) @ Wd @ expH(-1j * t)
        want = 2 * np.vdot(ref0, Wt @ Vd @ Wt @ Vd @ ref0).real + 0.5
        assert abs(got - want) < 1e-7
    finally:
        config.L = old


def test_tutorial_flows():
    """Code paths of the reference's tutorial notebooks (examples/tutorial/2-States, 3-Eigensolving,
    4-TimeEvolution, 5-Subspaces), checked against closed forms / dense algebra."""
    from dynamite_amd import config
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, index_sum, index_product, op_sum
    from dynamite_amd.subspaces import Auto
    oldL, olds = config.L, config.subspace
    try:
        # --- 2-States
        config.L = 6
        assert sigmaz(0).expectation(State(state='UUUUUU')) == 1 and sigmaz(0).expectation(State(state='DUUUUU')) == -1
        ghz = State(state='UUUUUU')
        ghz += State(state='DDDDDD')
        ghz.normalize()
        v = ghz.to_numpy()
        assert abs(v[0] - 2 ** -0.5) < 1e-15 and abs(v[-1] - 2 ** -0.5) < 1e-15 and np.count_nonzero(v) == 2
        w = State(state='100000')
        for i in range(1, w.L):
            w += State(state='0' * i + '1' + '0' * (w.L - i - 1))
        w.normalize()
        assert np.allclose(np.nonzero(w.to_numpy())[0], [1, 2, 4, 8, 16, 32]) and abs(w.norm() - 1) < 1e-15
        f = lambda st: np.exp(2 * np.pi * 1j * (st / 64))                       # noqa: E731
        s1, s2 = State(), State()
        s1.set_all_by_function(f)
        s2.set_all_by_function(f, vectorize=True)
        assert np.allclose(s1.to_numpy(), f(np.arange(64))) and np.array_equal(s1.to_numpy(), s2.to_numpy())
        assert abs(State(state='random', seed=1).norm() - 1) < 1e-14
        # --- 3-Eigensolving: transverse-field Ising ring
        config.L = 10
        H = 1 * index_sum(sigmaz(0) * sigmaz(1), boundary='closed') + 0.1 * index_sum(sigmax(0))
        dense = np.linalg.eigvalsh(H.to_numpy().toarray())
        assert abs(H.eigsolve()[0] - dense[0]) < 1e-7
        ev, vecs = H.eigsolve(nev=3, getvecs=True, tol=1e-10)
        assert len(ev) >= 3 and abs(H.expectation(vecs[0]) - ev[0]) < 1e-8
        mtot = index_sum(sigmaz(0))
        assert abs(mtot.expectation(vecs[0])) <= 10 + 1e-9
        # --- 4-TimeEvolution: domain wall melting, swapping state and result
        config.L = 10
        H = index_sum(sum(0.25 * p(0) * p(1) for p in [sigmax, sigmay, sigmaz]))
        Sz = [0.5 * sigmaz(i) for i in range(config.L)]
        cur = State(state='U' * 5 + 'D' * 5)
        res = cur.copy()
        tot0 = sum(S.expectation(cur) for S in Sz)
        for _ in range(5):
            H.evolve(cur, t=0.2, result=res)
            cur, res = res, cur
        w_, U = np.linalg.eigh(H.to_numpy().toarray())
        x0 = np.zeros(1 << 10, dtype=complex); x0[0b1111100000] = 1
        ref = U @ (np.exp(-1j * w_) * (U.conj().T @ x0))
        assert np.linalg.norm(cur.to_numpy() - ref) < 1e-7
        assert abs(sum(S.expectation(cur) for S in Sz) - tot0) < 1e-9        # total Sz conserved
        # --- 5-Subspaces
        config.L = 12
        H = index_sum(sum(0.25 * p(0) * p(1) for p in [sigmax, sigmay, sigmaz]))
        assert H.dim == (4096, 4096)
        H.subspace = SpinConserve(L=config.L, k=config.L // 2)
        assert H.dim == (924, 924)
        assert len(State(state='U' * 6 + 'D' * 6, subspace=SpinConserve(L=12, k=6))) == 924
        with pytest.raises(ValueError):
            State(state='U' * 5 + 'D' * 7, subspace=SpinConserve(L=12, k=6))
        for sector, eig in (('+', 1), ('-', -1)):
            config.subspace = XParity(SpinConserve(L=12, k=6), sector=sector)
            assert config.subspace.get_dimension() == 462
            flip = index_product(sigmax())
            psi = State(state='random', seed=2)
            assert abs(flip.expectation(psi) - eig) < 1e-12
        config.subspace = None
        XXZ = op_sum(index_sum(sigmax(0) * sigmax(i)) for i in range(1, 12)) + 0.5 * index_sum(sigmaz())
        XXZ.subspace = Parity('even')
        assert XXZ.dim == (2048, 2048)
        half = [x for x in range(1 << 12) if bin(x).count('1') == 6]
        assert Explicit(half) == SpinConserve(12, 6)
        heis = index_sum(sum(0.25 * p(0) * p(1) for p in [sigmax, sigmay, sigmaz]))
        assert Auto(heis, 'U' * 6 + 'D' * 6) == SpinConserve(12, 6)
        uns = Auto(heis, 'U' * 6 + 'D' * 6, sort=False)
        assert uns.get_dimension() == 924 and not np.array_equal(uns.state_map, np.sort(uns.state_map))
    finally:
        config.L, config.subspace = oldL, olds


@pytest.mark.parametrize("kind,P", [("full", 2), ("full", 8), ("even", 4), ("odd", 2)])
def test_rdm_partitioned_blocks(kind, P):
    """The partitioned reduced density matrix rank by rank on one GPU: every block's partial matrix on
    its block subspace (backend.rdm_block_subspace) summed over the ranks equals the matrix of the
    whole state; kept spins that reach the rank bits fall back to the gather."""
    from dynamite_amd.computations import reduced_density_matrix
    L = 12
    sub = {"full": Full(L=L), "even": Parity('even', L=L), "odd": Parity('odd', L=L)}[kind]
    st = State(L=L, subspace=sub, state='random', seed=6)
    nloc = len(st) // P
    p = P.bit_length() - 1
    # every rank's block in its own device layout (the swizzle acts on the local index)
    nat = st.to_numpy()
    blocks = [vec_from(nat[r * nloc:(r + 1) * nloc], sub.vec_swizzle) for r in range(P)]
    for keep in ([0], [1, 3, 4], list(range(6)), [2, L - p - 1]):
        keep = np.array(keep, dtype=np.int64)
        tot = 0
        for r in range(P):
            blk = backend.rdm_block_subspace(sub._to_c(), r, P, keep)
            assert blk is not None and blk.L == L - p
            tot = tot + backend.rdm_partial(blocks[r].array, blk, keep)
        K = 1 << keep.size
        assert np.max(np.abs(tot.cpu().numpy().reshape(K, K) - reduced_density_matrix(st, keep))) < 1e-14
    assert backend.rdm_block_subspace(sub._to_c(), 0, P, np.array([L - p])) is None
    assert backend.rdm_block_subspace(SpinConserve(L, 6)._to_c(), 0, P, np.array([0])) is None
    assert backend.rdm_block_subspace(sub._to_c(), 0, 3, np.array([0])) is None


@pytest.mark.skipif(not __import__("os").environ.get("DNM_TEST_MEMORY_PRESSURE"),
                    reason="fills the whole HBM (20 s); run with DNM_TEST_MEMORY_PRESSURE=1")
def test_workspace_released_on_memory_pressure():
    """The Krylov workspace cached after a solve is handed back when a later device allocation would
    otherwise fail.  (Verified on MI355X; opt-in because it allocates all of the device memory.)"""
    import ctypes as C
    import torch
    L = 24
    H = models.mbl(L)
    x = State(L=L, state='random', seed=0)
    H.evolve(x, t=0.1)
    cached = C.c_size_t()
    _lib.check(_lib.lib().dnm_workspace_bytes(C.byref(cached)))
    assert cached.value >= 30 * 16 * (1 << L)                 # 32 vectors of 256 MiB
    torch.cuda.empty_cache()                                  # nothing torch could recycle instead
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free - (100 << 20), dtype=torch.uint8, device='cuda')      # leave 100 MiB
    y = State(L=L)                                            # 256 MiB: needs the workspace back
    y.vec.set(1.0)
    _lib.check(_lib.lib().dnm_workspace_bytes(C.byref(cached)))
    assert cached.value == 0 and abs(y.vec.norm() - (1 << L) ** 0.5) < 1e-6
    del hog
    torch.cuda.empty_cache()
    H.evolve(x, t=0.1)                                        # and solves allocate it again


@pytest.mark.parametrize("real,getvecs", [("0", True), ("1", False), ("1", True)])
@pytest.mark.parametrize("name,L,sub,which", [("heisenberg", 12, "sc", "lowest"), ("mbl", 12, "full", "highest"),
                                              ("xxz", 11, "parity", "lowest"), ("long_range", 10, "full", "exterior")])
def test_eigsolve_deflated_pairs(monkeypatch, name, L, sub, which, real, getvecs):
    """Several pairs WITHOUT a stored basis (what eigsolve falls back to when a restarted basis does not fit in device
    memory -- the 36-site kagome torus; forced here): one pair after the other, the recurrence projected against the
    pairs found.  Values against the dense spectrum -- multiplicities included, which one Krylov space cannot see:
    the Heisenberg chain's levels are spin multiplets -- vectors by the reference's residual bar
    (tests/integration/test_eigsolve.py:17-88) and their mutual orthogonality."""
    monkeypatch.setenv("DNM_EIGS_BASISFREE", "1")
    monkeypatch.setenv("DNM_EIGS_REAL", real)
    for k, v in (("DNM_TILE_BITS", "8"), ("DNM_LOG_ROWS", "2"), ("DNM_PLAN_MODE", "2"), ("DNM_GBITS", "3"), ("DNM_AMIN", "3")):
        monkeypatch.setenv(k, v)          # small tiles: the real-arithmetic handle needs the tiled kernel
    H = models.BY_NAME[name](L)
    s = {"full": Full(L=L), "sc": SpinConserve(L, L // 2), "parity": Parity('even', L=L)}[sub]
    H.add_subspace(s)
    w = dense_spectrum(H, s)
    nev = 3
    if which == "lowest":
        want = w[:nev]
    elif which == "highest":
        want = w[::-1][:nev]
    else:
        want = w[np.argsort(-np.abs(w), kind="stable")][:nev]
    out = H.eigsolve(nev=nev, which=which, tol=1e-10, subspace=s, getvecs=getvecs)
    ev, vecs = out if getvecs else (out, [])
    from dynamite_amd.computations import eigsolve
    st = eigsolve.last_stats
    assert st['nconv'] == nev and st['max_rel_residual'] <= 1.01e-10
    if sub != "sc" and name != "long_range":
        # (SpinConserve has a real form in the internal layout only: test_gpu_sc3*.py; the packed form refuses the
        # long-range model under these small tiles and complex128 runs)
        assert st['real_arithmetic'] is (real == "1")
    assert len(ev) == nev and np.max(np.abs(np.array(ev) - want)) < 1e-8 * max(1.0, np.abs(want).max()), (ev, want)
    for i, (e, v) in enumerate(zip(ev, vecs)):
        r = H.dot(v)
        r.axpy(-e, v)
        assert r.norm() < 1e-8 * max(1.0, abs(e)) and abs(v.norm() - 1) < 1e-12
        for u in vecs[:i]:
            assert abs(u.dot(v)) < 1e-7
    H.destroy_mat()


def test_eigsolve_falls_back_to_deflation_when_memory_is_short(monkeypatch):
    """What computations.eigsolve does when no restarted basis fits (the 36-site kagome torus: five vectors of 34 GiB):
    device memory made to look short here -- six vectors' worth: the pairs come one after the other through the deflated
    basis-free recurrence (cap < nev + 6 in dnm_eigsolve); three vectors' worth: the error names what is needed."""
    import torch
    from dynamite_amd.computations import eigsolve
    L = 12
    H = models.mbl(L)
    w = dense_spectrum(H)
    vec_bytes = 16 << L
    cached = C.c_size_t()
    _lib.check(_lib.lib().dnm_release_workspace())
    total = torch.cuda.mem_get_info()[1]
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (6 * vec_bytes + 100, total))
    ev = H.eigsolve(nev=2, tol=1e-10)
    st = eigsolve.last_stats
    assert st["its"] == 2 and st["nconv"] == 2          # (the deflated driver reports one "restart" per pair)
    assert np.max(np.abs(np.array(ev[:2]) - w[:2])) < 1e-8
    _lib.check(_lib.lib().dnm_release_workspace())
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (3 * vec_bytes + 100, total))
    with pytest.raises(RuntimeError, match="basis-free solver needs"):
        H.eigsolve(nev=2, tol=1e-10)
    H.destroy_mat()


def test_eigsolve_deflated_degenerate_level():
    """The '-' sector of XParity(SpinConserve(12, 6)) on the 12-site kagome torus has a triply degenerate lowest
    level (tests/golden/kagome.npz, the reference's own reduced matrix): a single Krylov space holds one copy of it,
    the deflated recurrence returns all three and then the next level."""
    from dynamite_amd.subspaces import XParity
    import os as _os
    g = np.load(_os.path.join(_os.path.dirname(__file__), "golden", "kagome.npz"))
    want = g["kagome_12_sc_xparity_minus/evals_lowest"]
    assert abs(want[0] - want[2]) < 1e-10 and want[3] - want[2] > 1e-2
    H = models.kagome("12")
    sub = XParity(SpinConserve(12, 6), sector='-')
    H.add_subspace(sub)
    _os.environ["DNM_EIGS_BASISFREE"] = "1"
    try:
        ev = H.eigsolve(nev=4, tol=1e-11, subspace=sub)
    finally:
        _os.environ.pop("DNM_EIGS_BASISFREE")
    assert np.max(np.abs(np.array(ev[:4]) - want[:4])) < 1e-8, (ev, want[:4])
    H.destroy_mat()


def test_eigsolve_many_pairs():
    """nev large enough that SLEPc's default ncv = max(2 nev, nev + 15) exceeds 160 vectors (the basis rotation
    then stages fewer rows per step); the reference accepts any ncv."""
    L = 10
    H = models.mbl(L)
    ev = H.eigsolve(nev=90, tol=1e-9)
    want = np.linalg.eigvalsh(H.to_numpy(sparse=False))[:90]
    assert len(ev) >= 90 and np.max(np.abs(np.sort(ev)[:90] - want)) < 1e-7


@pytest.mark.parametrize("which", ["lowest", "highest", "exterior"])
@pytest.mark.parametrize("name,L,sub", [("mbl", 12, "full"), ("heisenberg", 12, "sc"), ("xxz", 11, "parity"),
                                        ("long_range", 10, "full")])
def test_eigsolve_basis_free(monkeypatch, name, L, sub, which):
    """The basis-free Lanczos path (one extremal pair; the default from 2^22 local amplitudes on, forced here):
    eigenvalue against dense diagonalisation, and with getvecs the reference's residual / Rayleigh-quotient bars
    (tests/integration/test_eigsolve.py:17-88)."""
    monkeypatch.setenv("DNM_EIGS_BASISFREE", "1")
    H = models.BY_NAME[name](L)
    s = {"full": Full(L=L), "sc": SpinConserve(L, L // 2), "parity": Parity('even', L=L)}[sub]
    H.add_subspace(s)
    dense = H.to_numpy(subspaces=(s, s), sparse=True).tocsr()
    w = dense_spectrum(H, s)
    want = {"lowest": w[0], "highest": w[-1], "exterior": w[0] if abs(w[0]) > abs(w[-1]) else w[-1]}[which]
    from dynamite_amd.computations import eigsolve
    ev = H.eigsolve(nev=1, which=which, tol=1e-10, subspace=s)
    assert len(ev) == 1 and abs(ev[0] - want) < 1e-8 * max(1.0, abs(want))
    assert eigsolve.last_stats['max_rel_residual'] < 1e-9
    ev2, vecs = H.eigsolve(nev=1, which=which, tol=1e-10, subspace=s, getvecs=True)
    v = vecs[0].to_numpy()
    assert abs(np.linalg.norm(v) - 1) < 1e-12
    assert abs(ev2[0] - want) < 1e-8 * max(1.0, abs(want))
    assert np.linalg.norm(dense @ v - ev2[0] * v) < 1e-8 * max(1.0, abs(want))      # measured, as promised
    assert eigsolve.last_stats['max_rel_residual'] < 1e-9


@pytest.mark.parametrize("which", ["lowest", "highest"])
@pytest.mark.parametrize("name,L,sub,nev", [("mbl", 12, "full", 5), ("heisenberg", 14, "sc", 3), ("xxz", 12, "parity", 4),
                                            ("long_range", 11, "full", 3)])
def test_eigsolve_filtered(monkeypatch, name, L, sub, nev, which):
    """Thick-restart Lanczos on a Chebyshev filter of H (several pairs at one end of the spectrum; the default from 2^22
    local amplitudes on, forced here): eigenvalues against dense diagonalisation -- every returned value must be an
    eigenvalue and the extremal one must be found (degenerate levels may be missed by a Krylov solver,
    computations.py:137-139) -- and the reference's Rayleigh-quotient / residual / orthogonality bars
    (tests/integration/test_eigsolve.py:17-88, 127-137) at tol = 1e-12."""
    monkeypatch.setenv("DNM_EIGS_FILTER", "1")
    H = models.BY_NAME[name](L)
    s = {"full": Full(L=L), "sc": SpinConserve(L, L // 2), "parity": Parity('even', L=L)}[sub]
    H.add_subspace(s)
    w = dense_spectrum(H, s)
    if which == "highest":
        w = w[::-1]
    from dynamite_amd.computations import eigsolve
    ev, vecs = H.eigsolve(nev=nev, which=which, tol=1e-12, subspace=s, getvecs=True)
    assert len(ev) >= nev
    assert abs(ev[0] - w[0]) < 1e-9 * max(1.0, abs(w[0]))
    for e in ev[:nev]:
        assert np.min(np.abs(w - e)) < 1e-9 * max(1.0, abs(e))
    assert all((ev[i] <= ev[i + 1] + 1e-12) if which == "lowest" else (ev[i] >= ev[i + 1] - 1e-12) for i in range(nev - 1))
    assert eigsolve.last_stats['max_rel_residual'] <= 1e-12
    for i, (e, v) in enumerate(zip(ev[:nev], vecs[:nev])):
        Hv = H.dot(v)
        assert abs(v.norm() - 1) < 1e-10
        assert abs(Hv.dot(v).real - e) < max(1e-12, abs(e) * 1e-12) * 10
        r = Hv.copy()
        r.axpy(-e, v)
        assert r.norm() < 1e-11 * max(1.0, abs(e))
        for j in range(i):
            assert abs(v.dot(vecs[j])) < 1e-12 * 10


def test_eigsolve_filtered_default_at_size():
    """L = 24 XXZ (2^24 amplitudes): eigsolve(nev=4) takes the filtered scheme on its own; its values agree with the
    plain restarted scheme and the residuals hold the requested tolerance."""
    import os
    from dynamite_amd.computations import eigsolve
    L = 24
    H = models.xxz(L)
    ev = H.eigsolve(nev=4, tol=1e-9)
    st = dict(eigsolve.last_stats)
    assert st['nconv'] >= 4 and st['max_rel_residual'] <= 1e-9
    os.environ["DNM_EIGS_FILTER"] = "0"
    try:
        ev0 = H.eigsolve(nev=4, tol=1e-9)
    finally:
        del os.environ["DNM_EIGS_FILTER"]
    assert np.abs(np.asarray(ev[:4]) - np.asarray(ev0[:4])).max() < 1e-7
    assert st['matvecs'] != eigsolve.last_stats['matvecs']          # two different schemes ran


@pytest.mark.default_layout
def test_vec_layout_conversions_chunked():
    """Vec index operations under the production layout on a block larger than the conversion chunk (2^24):
    numpy round trip (chunked both ways), single positions, ranges, the one-kernel conversion and the RNG stream
    (a swizzled vector holds the numbers index order would)."""
    import torch
    n = (1 << 25) + 0
    S = 16
    a = rand_state(n, seed=4)
    v = vec_from(a, S)
    assert v.swz == S
    back = v.local_numpy()
    assert np.array_equal(back, a)
    idx = torch.tensor([0, 15, 16, (1 << 16) - 1, 1 << 16, (1 << 16) + 17, (1 << 24) + 12345, n - 1], device=v.array.device)
    assert np.array_equal(v.array[v.positions(idx)].cpu().numpy(), a[idx.cpu().numpy()])
    assert not np.array_equal(v.array[idx].cpu().numpy(), a[idx.cpu().numpy()])      # the layout does permute
    lo, hi = (1 << 20) - 5, (1 << 20) + 70000
    assert np.array_equal(v.get_local(lo, hi).cpu().numpy(), a[lo:hi])
    nat = v.local_natural()
    assert np.array_equal(nat.cpu().numpy(), a)
    w = backend.Vec(n, swz=S)
    w.set_local(lo, hi, torch.from_numpy(a[lo:hi]).to(w.array.device))
    assert np.array_equal(w.get_local(lo, hi).cpu().numpy(), a[lo:hi]) and float(w.norm()) > 0
    r0, r1 = backend.Vec(n), backend.Vec(n, swz=S)
    r0.set_random(77)
    r1.set_random(77)
    assert np.array_equal(r1.local_numpy(), r0.local_numpy())


def test_evolve_probe_of_the_norm_bound(monkeypatch, capfd):
    """Memory-bound sizes: before acquiring a large Krylov workspace, evolve looks at the spectrum x sees (ten
    Lanczos steps) and takes the Chebyshev expansion unless the norm bound is loose -- forced on here at a small
    size (DNM_EXPM_PROBE): tight bound (random-field Heisenberg) -> expansion, loose bound (SYK) -> Krylov; results
    agree with scipy either way."""
    import scipy.sparse.linalg as spla
    from dynamite_amd import models
    from dynamite_amd.states import State
    monkeypatch.setenv("DNM_EXPM_PROBE", "1")
    monkeypatch.setenv("DNM_KRYLOV_DEBUG", "1")
    for name, L, t, want in (("mbl", 12, 3.0, "Chebyshev expansion"), ("syk", 8, 0.3, "Krylov")):
        H = models.BY_NAME[name](L)
        H.establish_L()
        x = State(L=H.L, state='random', seed=4)
        y = H.evolve(x, t=t)
        err = capfd.readouterr().err
        assert "spectral extent seen by x" in err and ("-> " + want) in err, err
        ref = spla.expm_multiply(-1j * t * H.to_numpy(sparse=True), x.to_numpy())
        assert np.max(np.abs(y.to_numpy() - ref)) < 1e-8


@pytest.mark.parametrize("name,L,sub", [("mbl", 16, "full"), ("ising", 15, "parity"), ("xxz", 14, "full"), ("xsum", 13, "full")])
def test_real_packed_multiply_vs_oracle(monkeypatch, name, L, sub):
    """DNM_MAT_REAL_PACKED on the GPU: the tiled kernel's real-arithmetic instance (two real amplitudes per element)
    against the oracle for a real x, the fused <x, y> / |y|^2 of the Lanczos step, and dnm_vec_unpack_real."""
    import ctypes as C
    from gpu_util import marshal, orc_msc, orc_sub, shell
    from oracle import oracle as orc
    from dynamite_amd import _lib
    for k, v in (("DNM_TILE_BITS", "8"), ("DNM_LOG_ROWS", "2"), ("DNM_PLAN_MODE", "2"), ("DNM_GBITS", "3"), ("DNM_AMIN", "3")):
        monkeypatch.setenv(k, v)
    H = models.BY_NAME[name](L)
    s = Full(L=L) if sub == "full" else Parity('even', L=L)
    dim = s.get_dimension()
    mat = shell(H, s, flags=_lib.MAT_REAL_PACKED)
    assert mat.N == dim // 2 and "tiled=1" in mat.describe()
    xv, yv = mat.createVecs()
    rs = np.random.RandomState(L)
    xr = rs.standard_normal(dim)
    xv.set_local_from_numpy(xr[0::2] + 1j * xr[1::2])
    yv.set(7.0)
    mat.mult(xv, yv)
    ref = orc.matvec(orc_msc(H), orc_sub(s), orc_sub(s), xr.astype(np.complex128)).real
    yp = yv.local_numpy()
    got = np.empty(dim)
    got[0::2], got[1::2] = yp.real, yp.imag
    tol = 64 * 2.2e-16 * np.abs(H.msc['coeffs']).sum() * np.abs(xr).max()
    assert np.abs(got - ref).max() <= tol
    # the Lanczos step's fused sums: <x, y> is the REAL inner product, |y|^2 the real norm
    d = (C.c_double * 3)()
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, xv.ptr, yv.ptr, None, 0.0, d, None))
    assert abs(d[0] - xr @ ref) <= 1e-11 * max(1.0, abs(xr @ ref)) and abs(d[2] - ref @ ref) <= 1e-11 * (ref @ ref)
    # unpacked: the complex vector of the full dimension, in the subspace's own layout
    from dynamite_amd import backend
    out = backend.Vec(dim, swz=s.vec_swizzle)
    _lib.check(_lib.lib().dnm_vec_unpack_real(out.ptr, yv.ptr, mat.n_local, mat.swz_right, s.vec_swizzle, None))
    assert np.abs(out.local_numpy() - ref).max() <= tol
    mat.destroy()


@pytest.mark.parametrize("name,L,sub,mode", [(n_, L_, s_, m_) for n_, L_, s_ in
                                             [("mbl", 12, "full"), ("xxz", 13, "parity"), ("heisenberg", 11, "full"),
                                              ("ising", 13, "fullx+"), ("heisenberg", 12, "fullx-")]
                                             for m_ in ("restarted", "basis_free", "filtered")
                                             # (one scheme for the second XParity sector: suite time)
                                             if not (s_ == "fullx-" and m_ != "restarted")])
def test_eigsolve_real_arithmetic(monkeypatch, name, L, sub, mode):
    """eigsolve of a real-symmetric operator in real arithmetic (the default from 2^23 amplitudes on one rank, forced
    here): the same eigenvalues as dense diagonalisation, and the returned COMPLEX states pass the reference's
    residual / Rayleigh-quotient / orthogonality bars (tests/integration/test_eigsolve.py:17-88, 127-137) -- through
    the restarted scheme, the basis-free Lanczos and the Chebyshev-filtered scheme."""
    from dynamite_amd.computations import eigsolve
    monkeypatch.setenv("DNM_EIGS_REAL", "1")
    for k, v in (("DNM_TILE_BITS", "8"), ("DNM_LOG_ROWS", "2"), ("DNM_PLAN_MODE", "2"), ("DNM_GBITS", "3"), ("DNM_AMIN", "3")):
        monkeypatch.setenv(k, v)          # small tiles: the real-arithmetic handle needs the tiled kernel
    nev = 1 if mode == "basis_free" else 3
    if mode == "basis_free":
        monkeypatch.setenv("DNM_EIGS_BASISFREE", "1")
    if mode == "filtered":
        monkeypatch.setenv("DNM_EIGS_FILTER", "1")
    H = models.BY_NAME[name](L)
    # (fullx: XParity on top of the Full space -- the Z2 sector of the transverse-field Ising chain, the spin-flip
    # sector of the Heisenberg chain: the reduced operator of subspaces.py:632-674 on the tiled kernel, packed)
    s = {"full": Full(L=L), "parity": Parity('even', L=L), "fullx+": XParity(Full(L=L), sector='+'),
         "fullx-": XParity(Full(L=L), sector='-')}[sub]
    H.add_subspace(s)
    w = dense_spectrum(H, s)
    ev, vecs = H.eigsolve(nev=nev, tol=1e-11, subspace=s, getvecs=True)
    assert eigsolve.last_stats['real_arithmetic'] is True
    assert len(ev) >= nev and abs(ev[0] - w[0]) < 1e-9 * max(1.0, abs(w[0]))
    for e in ev[:nev]:
        assert np.min(np.abs(w - e)) < 1e-9 * max(1.0, abs(e))
    for i, (e, v) in enumerate(zip(ev[:nev], vecs[:nev])):
        assert v.vec.size == s.get_dimension()
        Hv = H.dot(v)
        assert abs(v.norm() - 1) < 1e-10
        assert abs(Hv.dot(v).real - e) < 1e-10 * max(1.0, abs(e))
        r = Hv.copy()
        r.axpy(-e, v)
        assert r.norm() < 1e-9 * max(1.0, abs(e))
        assert np.abs(v.to_numpy().imag).max() == 0.0
        for j in range(i):
            assert abs(v.dot(vecs[j])) < 1e-10
    # values only, and the complex path on the same operator agrees
    ev_r = H.eigsolve(nev=nev, tol=1e-11, subspace=s)
    monkeypatch.setenv("DNM_EIGS_REAL", "0")
    ev_c = H.eigsolve(nev=nev, tol=1e-11, subspace=s)
    assert eigsolve.last_stats['real_arithmetic'] is False
    assert np.abs(np.asarray(ev_r[:nev]) - np.asarray(ev_c[:nev])).max() < 1e-9


def test_eigsolve_real_arithmetic_falls_back():
    """An operator with an imaginary matrix element (single sigma_y terms) keeps the complex path whatever is asked."""
    import os
    from dynamite_amd.computations import eigsolve
    os.environ["DNM_EIGS_REAL"] = "1"
    try:
        H = models.long_range(11)
        ev = H.eigsolve(nev=2, tol=1e-10)
        assert eigsolve.last_stats['real_arithmetic'] is False
        w = dense_spectrum(H)
        assert abs(ev[0] - w[0]) < 1e-8
    finally:
        del os.environ["DNM_EIGS_REAL"]


@pytest.mark.parametrize("which", ["highest", "exterior"])
def test_eigsolve_real_arithmetic_other_ends(monkeypatch, which):
    """The other ends of the spectrum in real arithmetic (restarted scheme and, for one pair, the basis-free Lanczos)."""
    from dynamite_amd.computations import eigsolve
    monkeypatch.setenv("DNM_EIGS_REAL", "1")
    for k, v in (("DNM_TILE_BITS", "8"), ("DNM_LOG_ROWS", "2"), ("DNM_PLAN_MODE", "2"), ("DNM_GBITS", "3"), ("DNM_AMIN", "3")):
        monkeypatch.setenv(k, v)
    L = 12
    H = models.mbl(L)
    w = dense_spectrum(H)
    want = w[-1] if which == "highest" else (w[0] if abs(w[0]) > abs(w[-1]) else w[-1])
    ev = H.eigsolve(nev=2, which=which, tol=1e-10)
    assert eigsolve.last_stats['real_arithmetic'] is True and abs(ev[0] - want) < 1e-8 * max(1.0, abs(want))
    monkeypatch.setenv("DNM_EIGS_BASISFREE", "1")
    ev1, v1 = H.eigsolve(nev=1, which=which, tol=1e-10, getvecs=True)
    assert eigsolve.last_stats['real_arithmetic'] is True and abs(ev1[0] - want) < 1e-8 * max(1.0, abs(want))
    r = H.dot(v1[0])
    r.axpy(-ev1[0], v1[0])
    assert r.norm() < 1e-8 * max(1.0, abs(want))
