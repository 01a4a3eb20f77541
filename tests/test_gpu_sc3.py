"""
SpinConserve state vectors in the three-field internal layout (dynamite_amd/csrc/sc3.h) and the two-pass multiply
that works in it -- against the oracle's MatMult_CPU_General restatement (bpetsc_template_2.c:371-412, index maps
bsubspace_impl.h:187-245) in reference order, against the reference-order kernels, and through the solvers.
Small sizes run the (a, w) = (6, 4) kernel instances; the production instances (14, 10) run from L = 25 on.
"""
import ctypes as C
import os

import numpy as np
import pytest

from dynamite_amd import _lib, backend, models
from dynamite_amd.config import config
from dynamite_amd.states import State
from dynamite_amd.subspaces import SpinConserve, Full, XParity
from gpu_util import shell, orc_msc, orc_sub, rand_state, vec_for, mult_numpy
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture
def small_layout():
    """(6, 4) field split for every SpinConserve subspace with L >= 11, whatever its dimension."""
    old = (config.sc_layout, config.sc_layout_min_dim)
    config.sc_layout, config.sc_layout_min_dim = (6, 4), 0
    yield
    config.sc_layout, config.sc_layout_min_dim = old


@pytest.fixture
def no_layout():
    old = config.sc_layout
    config.sc_layout = None
    yield
    config.sc_layout = old


def _dm_chain(L):
    """Hopping with a complex amplitude (XX + YY plus a Dzyaloshinskii-Moriya term XY - YX) and staggered fields:
    bond matrix elements that differ in the two directions and have imaginary parts."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, index_sum, op_sum
    hop = index_sum(0.3 * (sigmax(0) * sigmax(1) + sigmay(0) * sigmay(1)) +
                    0.2 * (sigmax(0) * sigmay(1) - sigmay(0) * sigmax(1)) + 0.15 * sigmaz(0) * sigmaz(1), size=L)
    H = hop + op_sum([(0.1 + 0.07 * i) * sigmaz(i) for i in range(L)])
    H.L = L
    return H


def _next_nearest(L):
    """Chain plus next-nearest-neighbour ZZ: more than one diagonal term that sees both Lo and the fields above it."""
    from dynamite_amd.operators import sigmaz, index_sum
    nnn = 0.3 * index_sum(sigmaz(0) * sigmaz(2), size=L)
    nnn.L = L
    return models.heisenberg(L) + nnn


MODELS = {"heisenberg": models.heisenberg, "mbl": models.mbl, "xxz": models.xxz, "dm": _dm_chain, "nnn": _next_nearest,
          "long_range": models.long_range}


def test_layout_round_trip(small_layout):
    """reference order -> internal -> reference order is the identity, the padding holds zeros, single positions
    agree with the bulk copy, and the internal length is what dnm_vec_layout_size says."""
    import torch
    for L, k in ((11, 5), (13, 6), (14, 3), (12, 12), (12, 0)):
        sub = SpinConserve(L, k)
        n = sub.get_dimension()
        v = vec_for(sub)
        assert v.internal and v.rows == n and v.local_size >= n
        x = rand_state(n, seed=L + k)
        v.set_local_from_numpy(x)
        assert np.array_equal(v.local_numpy(), x)
        pos = v.positions(torch.arange(n, device=v.array.device)).cpu().numpy()
        assert len(set(pos.tolist())) == n and pos.max() < v.local_size
        arr = v.array.cpu().numpy()
        assert np.array_equal(arr[pos], x)
        pad = np.ones(v.local_size, dtype=bool)
        pad[pos] = False
        assert not arr[pad].any()
        for i in (0, n // 2, n - 1):
            assert v.positions(int(i)) == pos[i]
        v.set(2.5)                       # VecSet must not leak into the padding (norms would see it)
        assert abs(v.norm() - 2.5 * np.sqrt(n)) < 1e-9 * np.sqrt(n)
        v.shift(1.0)
        assert abs(v.norm() - 3.5 * np.sqrt(n)) < 1e-9 * np.sqrt(n)


@pytest.mark.parametrize("name", sorted(MODELS))
@pytest.mark.parametrize("L,k", [(11, 5), (12, 6), (13, 4), (14, 7)])
def test_multiply_vs_oracle(small_layout, monkeypatch, name, L, k):
    """Every kernel path of the internal layout against the oracle: two tiled passes (chains; on-the-fly and cached
    diagonal; real symmetric and complex bonds; the long-range model, whose single-spin X / Y fields never act inside
    the subspace and whose all-to-all ZZ part needs the cached diagonal) and the row kernel."""
    H = MODELS[name](L)
    sub = SpinConserve(L, k)
    n = sub.get_dimension()
    x = rand_state(n, seed=3)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    tol = 64 * 2.2e-16 * (L + 2) * max(1.0, np.abs(H.msc['coeffs']).max()) * np.abs(x).max()
    for env in ({}, {"DNM_SC3_DIAG": "cached"}, {"DNM_SC3_TILED": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        mat = shell(H, sub)
        d = mat.describe()
        assert "internal layout" in d
        if env.get("DNM_SC3_TILED") == "0":
            assert "row kernel" in d
        else:
            assert "two-pass" in d
            assert ("real symmetric" in d) == (name in ("heisenberg", "mbl", "xxz", "nnn", "long_range"))
            assert ("diagonal cached" in d) == ("DNM_SC3_DIAG" in env or name == "long_range")
        if mat.uses_cached_diagonal():
            mat.precompute_diagonal()
        got = mult_numpy(mat, x)
        assert np.abs(got - want).max() <= tol, (name, env, d)
        mat.destroy()
        for k_ in env:
            monkeypatch.delenv(k_)


def test_cached_diagonal_in_reference_order(small_layout):
    """dnm_mat_get_diagonal hands the diagonal out row by row in reference order whatever the vectors' layout."""
    L, k = 12, 5
    H = models.mbl(L)
    sub = SpinConserve(L, k)
    mat = shell(H, sub)
    mat.precompute_diagonal()
    n = sub.get_dimension()
    got = np.empty(n)
    _lib.check(_lib.lib().dnm_mat_get_diagonal(mat.handle, _lib.pf64(got), backend._stream()))
    want = orc.precompute_diagonal(orc_msc(H), orc_sub(sub))
    assert np.abs(got - want).max() < 1e-13
    mat.destroy()


def test_fused_sums_and_start_vectors(small_layout):
    """dnm_mat_mult_lanczos (y = A x - b z with <x,y>, |y|^2 from the last pass) and dnm_mat_mult_sub2
    (y = A x - b z + c z2) against the unfused sequence."""
    L, k = 13, 6
    sub = SpinConserve(L, k)
    H = _dm_chain(L)
    mat = shell(H, sub)
    x, y, z, z2, ref = (vec_for(sub) for _ in range(5))
    x.set_random(1); z.set_random(2); z2.set_random(3)
    b, c = 0.37, complex(-0.4, 0.9)
    mat.mult(x, ref)
    ref.axpby(-b, 1.0, z)
    dot = (C.c_double * 3)()
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, x.ptr, y.ptr, z.ptr, b, dot, backend._stream()))
    d_ref = ref.dot(x)            # sum ref_i conj(x_i) = <x, ref>
    assert abs(complex(dot[0], dot[1]) - d_ref) < 1e-10 * abs(d_ref)
    assert abs(dot[2] - ref.norm() ** 2) < 1e-10 * dot[2]
    y.axpby(-1.0, 1.0, ref)
    assert y.norm() < 1e-12 * ref.norm()
    _lib.check(_lib.lib().dnm_mat_mult_sub2(mat.handle, x.ptr, y.ptr, z.ptr, b, z2.ptr, c.real, c.imag, backend._stream()))
    ref.axpby(c, 1.0, z2)
    y.axpby(-1.0, 1.0, ref)
    assert y.norm() < 1e-12 * ref.norm()
    mat.destroy()


def test_states_and_solvers_in_the_layout(small_layout):
    """The State surface (product / random / project / to_numpy / files / reduced density matrix) and evolve /
    eigsolve with internal-layout vectors against reference-order runs of the same things."""
    L, k = 12, 6
    sub = SpinConserve(L, k)
    H = models.mbl(L)
    H.add_subspace(sub)
    s = State(L=L, subspace=sub, state='random', seed=5)
    assert s.vec.internal
    x = s.to_numpy()
    assert abs(np.linalg.norm(x) - 1.0) < 1e-12
    # the reference stream of set_random (host-generated for small states) lands in reference order
    R = np.random.RandomState(5)
    n = sub.get_dimension()
    xr = R.standard_normal(n) + 1j * R.standard_normal(n)
    assert np.abs(x - xr / np.linalg.norm(xr)).max() < 1e-14
    p = State(L=L, subspace=sub, state='U' * (L - k) + 'D' * k)
    idx = sub.state_to_idx(State.str_to_state('U' * (L - k) + 'D' * k, L))
    e = np.zeros(n, dtype=np.complex128); e[idx] = 1
    assert np.array_equal(p.to_numpy(), e)
    # project, against numpy
    q = s.copy()
    q.project(3, 1)
    sts = sub.idx_to_state(np.arange(n))
    xq = np.where(((sts >> 3) & 1) == 1, x, 0)
    assert np.abs(q.to_numpy() - xq / np.linalg.norm(xq)).max() < 1e-13
    # reduced density matrix
    from dynamite_amd.computations import reduced_density_matrix
    rho = reduced_density_matrix(s, [0, 1, 4])
    full = np.zeros(1 << L, dtype=np.complex128); full[sts] = x
    psi = full.reshape([2] * L)            # axis j <-> spin L-1-j
    keep_axes = [L - 1 - i for i in (4, 1, 0)]
    m = np.moveaxis(psi, keep_axes, [0, 1, 2]).reshape(8, -1)
    assert np.abs(rho - m @ m.conj().T).max() < 1e-13
    # evolve and eigsolve against scipy on the oracle's matrix-free product
    y = H.evolve(s, t=0.6)
    osub = orc_sub(sub)
    msc = orc_msc(H)
    eye = np.eye(n, dtype=np.complex128)
    dense = np.column_stack([orc.matvec(msc, osub, osub, np.ascontiguousarray(eye[:, j])) for j in range(n)])
    w, V = np.linalg.eigh(dense)
    want = V @ (np.exp(-0.6j * w) * (V.conj().T @ x))
    assert abs(1 - np.vdot(want, y.to_numpy()) / np.vdot(want, want)) < 1e-9
    vals, vecs = H.eigsolve(nev=3, getvecs=True, tol=1e-11, subspace=sub)
    assert np.abs(vals[:3] - w[:3]).max() < 1e-9
    for ev, v in zip(vals[:3], vecs[:3]):
        assert v.vec.internal
        r = dense @ v.to_numpy() - ev * v.to_numpy()
        assert np.linalg.norm(r) < 1e-8
    # files
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        s.save(os.path.join(d, 'st'))
        t = State.from_file(os.path.join(d, 'st'))
        assert np.array_equal(t.to_numpy(), x)


def test_projection_pairs_convert(small_layout):
    """An operator between the SpinConserve subspace and another one works in reference order: vectors in the internal
    layout are converted on the way in and out."""
    L, k = 12, 6
    sub, full = SpinConserve(L, k), Full(L=L)
    H = models.long_range(L)
    x = rand_state(sub.get_dimension(), seed=9)
    mat = shell(H, full, sub)                    # Full <- SpinConserve
    assert mat.swz_right == 0
    xv = vec_for(sub)
    xv.set_local_from_numpy(x)
    yv = vec_for(full)
    mat.mult(xv, yv)
    want = orc.matvec(orc_msc(H), orc_sub(full), orc_sub(sub), x)
    got = yv.local_numpy()
    assert np.abs(got - want).max() < 1e-12
    mat2 = shell(H, sub, full)                   # SpinConserve <- Full
    y2 = vec_for(sub)
    y2.set(5.0)
    mat2.mult(yv, y2)
    want2 = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(full), got)
    assert np.abs(y2.local_numpy() - want2).max() < 1e-11
    mat.destroy(); mat2.destroy()
    # XParity on top of the subspace: the first half of the layout (tests/test_gpu_sc3_graph.py), or reference order
    xp = XParity(SpinConserve(L, k), '+')
    v = State(L=L, subspace=xp, state='random', seed=1).vec
    assert xp.vec_swizzle == sub.vec_swizzle and v.internal and v.half and v.rows == sub.get_dimension() // 2
    config.sc_xparity_layout = False
    try:
        assert xp.vec_swizzle == 0 and not State(L=L, subspace=xp, state='random', seed=1).vec.internal
    finally:
        config.sc_xparity_layout = True


@pytest.mark.parametrize("L,k,name", [(25, 12, "mbl"), (26, 13, "dm"), (26, 11, "heisenberg")])
def test_production_instances_against_reference_order(L, k, name):
    """(a, w) = (14, 10), the instances large subspaces get: the two passes against the reference-order block kernel
    element-wise, Hermiticity, and the fused sums."""
    H = MODELS[name](L)
    sub = SpinConserve(L, k)
    assert sub.vec_swizzle == (14 | (10 << 8))
    mat = shell(H, sub)
    assert "two-pass" in mat.describe()
    a, b, Ha, Hb = (vec_for(sub) for _ in range(4))
    a.set_random(1); b.set_random(2)
    a.normalize(); b.normalize()
    mat.mult(a, Ha)
    mat.mult(b, Hb)
    assert abs(Hb.dot(a) - b.dot(Ha)) < 1e-11          # <a, H b> = <H a, b>
    old = config.sc_layout
    config.sc_layout = None
    try:
        subn = SpinConserve(L, k)
        matn = shell(H, subn)
        assert "internal layout" not in matn.describe()
        an, Hn = vec_for(subn), vec_for(subn)
        an.array.copy_(a.local_natural())
        matn.mult(an, Hn)
        diff = (Ha.local_natural() - Hn.array).abs().max().item()
        assert diff < 1e-13
        matn.destroy()
    finally:
        config.sc_layout = old
    dot = (C.c_double * 3)()
    y = vec_for(sub)
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, a.ptr, y.ptr, b.ptr, 0.25, dot, backend._stream()))
    Ha.axpby(-0.25, 1.0, b)
    assert abs(complex(dot[0], dot[1]) - Ha.dot(a)) < 1e-11 and abs(dot[2] - Ha.norm() ** 2) < 1e-11
    y.axpby(-1.0, 1.0, Ha)
    assert y.norm() < 1e-13
    mat.destroy()


def _random_chain(L, rs):
    """Nearest-neighbour chain with random complex hopping per bond (some bonds missing), random ZZ couplings at
    distance 1..3 and random fields: exercises absent bonds, direction-dependent complex matrix elements and several
    diagonal patterns that see both Lo and the fields above it."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum
    terms = []
    for i in range(L - 1):
        if rs.rand() < 0.2:
            continue                                     # no hopping on this bond
        a, b = rs.uniform(-1, 1), rs.uniform(-1, 1)
        terms.append(a * (sigmax(i) * sigmax(i + 1) + sigmay(i) * sigmay(i + 1)))
        if rs.rand() < 0.6:
            terms.append(b * (sigmax(i) * sigmay(i + 1) - sigmay(i) * sigmax(i + 1)))
    for i in range(L):
        terms.append(rs.uniform(-1, 1) * sigmaz(i))
        for dist in (1, 2, 3):
            if i + dist < L and rs.rand() < 0.5:
                terms.append(rs.uniform(-1, 1) * sigmaz(i) * sigmaz(i + dist))
    H = op_sum(terms)
    H.L = L
    return H


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_SC3_FUZZ_N", "64"))))
def test_fuzz_internal_layout(small_layout, seed):
    """Random chains (two tiled passes) and random Pauli-string sums (row kernel) on random SpinConserve sectors in the
    internal layout, against the oracle."""
    from test_gpu_matvec import _random_hermitian
    rs = np.random.RandomState(1000 + seed)
    L = int(rs.randint(11, 15))
    k = int(rs.randint(1, L))
    H = _random_chain(L, rs) if seed % 3 else _random_hermitian(L, 12, rs)
    sub = SpinConserve(L, k)
    n = sub.get_dimension()
    x = rand_state(n, seed=seed)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    mat = shell(H, sub)
    assert "internal layout" in mat.describe()
    if seed % 3:
        assert "two-pass" in mat.describe()
    if mat.uses_cached_diagonal():
        mat.precompute_diagonal()
    got = mult_numpy(mat, x)
    scale = max(1.0, np.abs(H.msc['coeffs']).sum()) * np.abs(x).max()
    assert np.abs(got - want).max() <= 64 * 2.2e-16 * scale, (L, k, mat.describe())
    mat.destroy()


def _real_mult(mat, sub, xr):
    """y = A x for a real x through a DNM_MAT_REAL_PACKED handle of a SpinConserve pair: the vectors are one double per
    position of the internal layout (nint / 2 complex128 elements); returns y in reference order."""
    import torch
    v = vec_for(sub)
    v.set_local_from_numpy(xr.astype(np.complex128))
    xd = v.array.real.contiguous()                      # double[nint], padding zero
    assert xd.numel() == v.local_size and mat.n_local * 2 == v.local_size
    yd = torch.full_like(xd, 7.0)
    _lib.check(_lib.lib().dnm_mat_mult(mat.handle, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None))
    torch.cuda.synchronize()
    out = vec_for(sub)
    _lib.check(_lib.lib().dnm_vec_layout_unpack_real(C.byref(sub._c()), None, out.ptr, C.c_void_p(yd.data_ptr()), None))
    pad = np.ones(v.local_size, dtype=bool)
    pad[v.positions(torch.arange(sub.get_dimension(), device=v.array.device)).cpu().numpy()] = False
    assert np.all(yd.cpu().numpy()[pad] == 0.0), "padding of the result is not zero"
    return out.local_numpy(), xd, yd


@pytest.mark.parametrize("name,L,k", [(n_, L_, k_) for L_, k_ in [(11, 5), (13, 6), (14, 3), (14, 7), (12, 11), (24, 12)]
                                      for n_ in ("heisenberg", "mbl", "xxz", "nnn")
                                      if L_ != 24 or n_ == "mbl"])       # (one operator at the larger size)
def test_real_packed_spinconserve_multiply(small_layout, name, L, k):
    """The two tiled passes on REAL vectors (sc3_lo_pass_r; the window pass on the halved tables) against the oracle,
    with the fused sums of the Lanczos step."""
    H = MODELS[name](L)
    sub = SpinConserve(L, k)
    n = sub.get_dimension()
    mat = shell(H, sub, flags=_lib.MAT_REAL_PACKED)
    assert "two-pass" in mat.describe()
    xr = np.random.RandomState(L * 31 + k).standard_normal(n)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xr.astype(np.complex128), nthreads=4)
    assert np.abs(want.imag).max() == 0.0
    got, xd, yd = _real_mult(mat, sub, xr)
    scale = max(1.0, np.abs(H.msc['coeffs']).sum()) * np.abs(xr).max()
    assert np.abs(got.imag).max() == 0.0
    assert np.abs(got.real - want.real).max() <= 64 * 2.2e-16 * scale, mat.describe()
    d = (C.c_double * 3)()
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None,
                                               0.0, d, None))
    assert abs(d[0] - xr @ want.real) <= 1e-11 * max(1.0, abs(xr @ want.real))
    assert abs(d[2] - want.real @ want.real) <= 1e-11 * (want.real @ want.real)
    mat.destroy()


def _random_real_chain(L, rs):
    """Nearest-neighbour chain with a random REAL hopping per bond (some bonds missing), random ZZ couplings at distance
    1..3 and random fields: real symmetric, a chain -- what the real two-pass form accepts."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum
    terms = []
    for i in range(L - 1):
        if rs.rand() < 0.2:
            continue
        terms.append(rs.uniform(-1, 1) * (sigmax(i) * sigmax(i + 1) + sigmay(i) * sigmay(i + 1)))
    for i in range(L):
        terms.append(rs.uniform(-1, 1) * sigmaz(i))
        for dist in (1, 2, 3):
            if i + dist < L and rs.rand() < 0.5:
                terms.append(rs.uniform(-1, 1) * sigmaz(i) * sigmaz(i + dist))
    H = op_sum(terms)
    H.L = L
    return H


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_SC3_FUZZ_REAL_N", "32"))))
def test_fuzz_internal_layout_real(small_layout, seed):
    """The real-arithmetic form of the two tiled passes on random real chains and random SpinConserve sectors, with its
    fused sums, against the oracle."""
    rs = np.random.RandomState(7000 + seed)
    L = int(rs.randint(11, 16))
    k = int(rs.randint(1, L))
    H = _random_real_chain(L, rs)
    sub = SpinConserve(L, k)
    n = sub.get_dimension()
    try:
        # (raw vectors of the subspace's own layout below: no site relabelling -- a chain with missing bonds may get one)
        mat = shell(H, sub, flags=_lib.MAT_REAL_PACKED, site_perm=False)
    except _lib.BackendError as e:          # (more mixed diagonal patterns than the on-the-fly diagonal takes: no real form)
        assert "real-packed" in str(e)
        return
    xr = rs.standard_normal(n)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xr.astype(np.complex128))
    assert np.abs(want.imag).max() == 0.0
    got, xd, yd = _real_mult(mat, sub, xr)
    scale = max(1.0, np.abs(H.msc['coeffs']).sum()) * np.abs(xr).max()
    assert np.abs(got.imag).max() == 0.0
    assert np.abs(got.real - want.real).max() <= 64 * 2.2e-16 * scale, (L, k, mat.describe())
    d = (C.c_double * 3)()
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None,
                                               0.0, d, None))
    assert abs(d[0] - xr @ want.real) <= 1e-11 * max(1.0, abs(xr @ want.real))
    assert abs(d[2] - want.real @ want.real) <= 1e-11 * max(1e-300, want.real @ want.real)
    mat.destroy()


def test_real_packed_spinconserve_refusals(small_layout):
    """No real form for an operator with imaginary bond elements.  (The long-range model -- a chain with an all-to-all
    diagonal -- has one since round 5: its real lo pass reads the cached diagonal.)"""
    sub = SpinConserve(13, 6)
    with pytest.raises(_lib.BackendError):
        shell(_dm_chain(13), sub, flags=_lib.MAT_REAL_PACKED)
    H = models.long_range(13)
    mat = shell(H, sub, flags=_lib.MAT_REAL_PACKED, site_perm=False)
    assert "diagonal cached" in mat.describe()
    mat.precompute_diagonal()
    xr = np.random.RandomState(4).standard_normal(sub.get_dimension())
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xr.astype(np.complex128))
    got, _, _ = _real_mult(mat, sub, xr)
    assert np.abs(got.imag).max() == 0.0
    assert np.abs(got.real - want.real).max() <= 64 * 2.2e-16 * np.abs(H.msc['coeffs']).sum() * np.abs(xr).max()
    mat.destroy()


@pytest.mark.default_layout
@pytest.mark.parametrize("L,k", [(25, 12), (26, 13)])
def test_real_packed_spinconserve_production_instance(L, k):
    """The (14, 10) instance (1024 threads, eight entries per thread, 2^m rows per workgroup) against the oracle."""
    sub = SpinConserve(L, k)
    assert sub.vec_swizzle == (14 | (10 << 8))
    H = models.mbl(L)
    mat = shell(H, sub, flags=_lib.MAT_REAL_PACKED)
    n = sub.get_dimension()
    xr = np.random.RandomState(L).standard_normal(n)
    want = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xr.astype(np.complex128), nthreads=min(16, orc.max_threads()))
    got, _, _ = _real_mult(mat, sub, xr)
    scale = max(1.0, np.abs(H.msc['coeffs']).sum()) * np.abs(xr).max()
    assert np.abs(got.real - want.real).max() <= 64 * 2.2e-16 * scale
    mat.destroy()


@pytest.mark.parametrize("mode", ["restarted", "basis_free", "filtered"])
def test_eigsolve_real_arithmetic_spinconserve(monkeypatch, small_layout, mode):
    """eigsolve in real arithmetic on a SpinConserve subspace in the internal layout: eigenvalues against dense
    diagonalisation, the returned complex states against the reference's residual / orthogonality bars."""
    from dynamite_amd.computations import eigsolve
    monkeypatch.setenv("DNM_EIGS_REAL", "1")
    if mode == "basis_free":
        monkeypatch.setenv("DNM_EIGS_BASISFREE", "1")
    if mode == "filtered":
        monkeypatch.setenv("DNM_EIGS_FILTER", "1")
    nev = 1 if mode == "basis_free" else 3
    L, k = 14, 7
    H = models.mbl(L)
    sub = SpinConserve(L, k)
    H.add_subspace(sub)
    w = np.linalg.eigvalsh(H.to_numpy(subspaces=(sub, sub), sparse=False))
    ev, vecs = H.eigsolve(nev=nev, tol=1e-11, subspace=sub, getvecs=True)
    assert eigsolve.last_stats['real_arithmetic'] is True
    for e in ev[:nev]:
        assert np.min(np.abs(w - e)) < 1e-9 * max(1.0, abs(e))
    assert abs(ev[0] - w[0]) < 1e-9 * max(1.0, abs(w[0]))
    for i, (e, v) in enumerate(zip(ev[:nev], vecs[:nev])):
        assert v.vec.internal and v.vec.rows == sub.get_dimension()
        Hv = H.dot(v)
        assert abs(v.norm() - 1) < 1e-10
        r = Hv.copy()
        r.axpy(-e, v)
        assert r.norm() < 1e-9 * max(1.0, abs(e))
        assert np.abs(v.to_numpy().imag).max() == 0.0
        for j in range(i):
            assert abs(v.dot(vecs[j])) < 1e-10


def _order1(sub):
    """descriptor dict of ``sub`` with its T blocks in the order made for partitions (vec_swizzle bits 16-19 = 1)"""
    desc = sub._to_c()
    d = type(desc['data']).from_buffer_copy(desc['data'])
    d.vec_swizzle = int(d.vec_swizzle) | (1 << 16)
    return {'type': desc['type'], 'data': d, '_keep': desc}


@pytest.mark.parametrize("name,L,k", [("heisenberg", 13, 6), ("mbl", 14, 7), ("dm", 14, 5), ("nnn", 15, 7), ("long_range", 13, 6)])
def test_block_order_for_partitions_on_one_rank(small_layout, name, L, k):
    """The layout with its T blocks in the order made for partitions (csrc/sc3.h: sc3_code_order 1) is the same layout to
    every kernel -- blocks are found through the ibase table wherever they lie: multiply against the oracle through the
    whole-vector maps to and from the reference order, positions a bijection that differs from the reference-compatible
    order's, seeded random vectors hold the same numbers, real arithmetic alike."""
    import torch
    from gpu_util import marshal
    H = MODELS[name](L)
    sub = SpinConserve(L, k)
    n = sub.get_dimension()
    sd = _order1(sub)
    code = int(sd['data'].vec_swizzle)
    assert code >> 16 == 1 and (code & 0xffff) == sub.vec_swizzle
    mat = backend.build_mat(*marshal(H), sd, sd, site_perm=False)
    if mat.uses_cached_diagonal():
        mat.precompute_diagonal()                 # (one rank: the whole vector has a reference side)
    assert "internal layout" in mat.describe() and mat.swz_right == code
    x = rand_state(n, seed=L)
    xv, yv = mat.createVecs()
    assert xv.swz == code and xv.local_size == vec_for(sub).local_size
    xv.set_local_from_numpy(x)
    assert np.array_equal(xv.local_numpy(), x)
    p1 = xv.positions(torch.arange(n, device=xv.array.device)).cpu().numpy()
    p0 = vec_for(sub).positions(torch.arange(n, device=xv.array.device)).cpu().numpy()
    assert len(set(p1.tolist())) == n and p1.max() < xv.local_size
    assert (L - 10 < 4) or not np.array_equal(p0, p1)       # (eight T blocks or fewer lie in ascending order either way)
    mat.mult(xv, yv)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    assert np.abs(yv.local_numpy() - ref).max() <= 64 * 2.2e-16 * np.abs(H.msc['coeffs']).sum() * np.abs(x).max()
    # a seeded random vector: the numbers the reference order gets, whatever the block order
    r1, r0 = mat.createVecs()[0], vec_for(sub)
    r1.set_random(5)
    r0.set_random(5)
    assert np.array_equal(r1.local_numpy(), r0.local_numpy())
    mat.destroy()


def test_block_order_for_partitions_production_instance():
    """... and the (14, 10) instances at SpinConserve(26, 13) (10.4 M states, T = 2 bits: four blocks in the order
    0, 1, 2, 3 -> 0, 1 | 2, 3 by ones above the lowest bit): multiply against the oracle, complex and real arithmetic."""
    from gpu_util import marshal
    L, k = 26, 13
    H = models.heisenberg(L)
    sub = SpinConserve(L, k)
    assert sub.vec_swizzle == (14 | (10 << 8))
    sd = _order1(sub)
    x = rand_state(sub.get_dimension(), seed=3)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=8)
    mat = backend.build_mat(*marshal(H), sd, sd, site_perm=False)
    xv, yv = mat.createVecs()
    xv.set_local_from_numpy(x)
    mat.mult(xv, yv)
    assert np.abs(yv.local_numpy() - ref).max() < 1e-12
    mat.destroy()
    pm = backend.build_mat(*marshal(H), sd, sd, flags=_lib.MAT_REAL_PACKED, site_perm=False)
    assert pm.real_packed
    xr = backend.Vec(sub.get_dimension(), swz=pm.swz_right, sub_c=sd['data'])
    xr.set_local_from_numpy(x.real + 0j)
    # pack: the real parts of the layout's positions, one double each
    import torch
    xp = torch.view_as_real(xr.array)[:, 0].contiguous()
    yp = torch.zeros_like(xp)
    xq = backend.RawVec(xp.view(torch.complex128), pm.swz_right)
    yq = backend.RawVec(yp.view(torch.complex128), pm.swz_left)
    pm.mult(xq, yq)
    yfull = backend.Vec(sub.get_dimension(), swz=pm.swz_right, sub_c=sd['data'])
    torch.view_as_real(yfull.array)[:, 0] = yp
    refr = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x.real + 0j, nthreads=8)
    assert np.abs(yfull.local_numpy() - refr).max() < 1e-12
    pm.destroy()
