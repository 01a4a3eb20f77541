"""
Parity tests proper: the HIP path (through the C ABI) against the golden
fixtures, against the CPU oracle on seeded inputs, and -- at the full
BASELINE sizes -- through size-independent properties.
Tolerance (reference's own, tests/integration/test_multiply.py:194-195):
|dy| <= nnz * 2.2e-16 per unit |coeff| and |x| (scaled by the data magnitudes).
"""
import numpy as np
import pytest

from dynamite_amd import _lib, models, backend
from dynamite_amd.subspaces import Full, Parity, SpinConserve, Explicit
from oracle import oracle as orc
from gpu_util import partner_slice, marshal, orc_msc, orc_sub, shell, vec_from, mult_numpy, rand_state

pytestmark = pytest.mark.gpu
EPS = 2.2e-16


def tol_for(arrs, x):
    return 8 * len(arrs[0]) * EPS * max(1.0, np.abs(arrs[3]).max()) * max(1.0, np.abs(x).max())


def cfg(monkeypatch, B=12, logR=4, mode=0, amin=3, gbits=6):
    monkeypatch.setenv("DNM_GBITS", str(gbits))
    monkeypatch.setenv("DNM_TILE_BITS", str(B))
    monkeypatch.setenv("DNM_LOG_ROWS", str(logR))
    monkeypatch.setenv("DNM_PLAN_MODE", str(mode))
    monkeypatch.setenv("DNM_AMIN", str(amin))


GOLD = [('mbl', 6), ('mbl', 10), ('mbl', 12), ('heisenberg', 10), ('xxz', 10), ('ising', 10),
        ('long_range', 8), ('localized', 10), ('syk', 5), ('xsum', 8)]


@pytest.mark.parametrize("flags", [_lib.MAT_FORCE_GATHER, 0, _lib.MAT_USE_GLDS])
@pytest.mark.parametrize("name,L", GOLD)
def test_golden_full_space(monkeypatch, golden_full, name, L, flags):
    """y = Hx, ||H||_inf and the diagonal against vectors derived from the
    reference's msc_to_numpy."""
    cfg(monkeypatch, B=8, logR=2)
    g = golden_full[f"{name}_L{L}"]
    H = models.BY_NAME[name](L)
    arrs = marshal(H)
    sub = Full(L=L)
    mat = shell(H, sub, flags=flags)
    if L >= 8 and not (flags & _lib.MAT_FORCE_GATHER):
        assert "tiled=1" in mat.describe()
    y = mult_numpy(mat, g["x"])
    assert np.max(np.abs(y - g["y"])) <= tol_for(arrs, g["x"]), mat.describe()
    nrm = mat.norm()
    assert abs(nrm - float(g["infnorm"])) <= len(arrs[0]) * EPS * 100 * max(1.0, nrm)
    if arrs[0][0] == 0:
        mat.precompute_diagonal()
        d = np.empty(1 << L)
        _lib.check(_lib.lib().dnm_mat_get_diagonal(mat.handle, _lib.pf64(d), None))
        assert np.max(np.abs(d - g["diag"].real)) <= 64 * EPS * np.abs(arrs[3]).sum()
        y = mult_numpy(mat, g["x"])      # generic kernel now uses the cached diagonal
        assert np.max(np.abs(y - g["y"])) <= tol_for(arrs, g["x"])
    mat.destroy()


def _subs_for(name, L, gs):
    p = name.split("_")
    def one(tag):
        if tag.startswith("sc"):
            return SpinConserve(L, int(tag[2:]))
        return {"even": Parity(0, L=L), "odd": Parity(1, L=L), "full": Full(L=L)}[tag]
    if "explicit" in name:
        st = gs["explicit_states"]["unsorted" if name.endswith("unsorted") else "sorted"]
        s = Explicit(st, L=L)
        return s, s
    if "_to_" in name:
        i = p.index("to")
        return one(p[i + 1]), one(p[i - 1])
    s = one(p[-1])
    return s, s


def test_golden_subspaces(monkeypatch, golden_sub):
    """SpinConserve / Parity / Explicit and the projection pairs (left != right)."""
    cfg(monkeypatch, B=8, logR=2)
    for name in golden_sub.names():
        if name == "explicit_states":
            continue
        g = golden_sub[name]
        L = int(g["L"])
        hname = next(k for k in models.BY_NAME if name.startswith(k))
        H = models.BY_NAME[hname](L)
        arrs = marshal(H)
        left, right = _subs_for(name, L, golden_sub)
        for flags in (0, _lib.MAT_FORCE_GATHER):
            mat = shell(H, left, right, flags=flags)
            y = mult_numpy(mat, g["x"])
            assert y.shape == g["y"].shape
            assert np.max(np.abs(y - g["y"])) <= tol_for(arrs, g["x"]), (name, flags)
            nrm = mat.norm()
            assert abs(nrm - float(g["infnorm"])) <= len(arrs[0]) * EPS * 100 * max(1.0, nrm), name
            mat.destroy()


CONFIGS = [(12, 4, 0, 3), (12, 3, 0, 3), (13, 4, 0, 3), (13, 3, 0, 4), (11, 4, 0, 3), (10, 3, 0, 5),
           (12, 4, 1, 3), (13, 4, 1, 3), (12, 3, 2, 4), (13, 3, 2, 4), (11, 3, 2, 3), (10, 4, 2, 4)]


@pytest.mark.parametrize("B,logR,mode,amin", CONFIGS)
@pytest.mark.parametrize("name,L", [("mbl", 20), ("long_range", 16), ("syk", 7)])
def test_tiled_vs_oracle(monkeypatch, name, L, B, logR, mode, amin):
    """Every tile configuration / plan mode against the CPU oracle, both tile
    staging paths."""
    cfg(monkeypatch, B, logR, mode, amin)
    H = models.BY_NAME[name](L)
    arrs = marshal(H)
    sub = Full(L=L)
    x = rand_state(1 << L, seed=L)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=4)
    for flags in (0, _lib.MAT_USE_GLDS):
        mat = shell(H, sub, flags=flags)
        assert ("tiled=1" in mat.describe()) == (L >= B)
        y = mult_numpy(mat, x)
        assert np.max(np.abs(y - ref)) <= tol_for(arrs, x), mat.describe()
        mat.destroy()


@pytest.mark.parametrize("spaces", [(0, 0), (1, 1), (0, 1)])
def test_parity_tiled_vs_oracle(monkeypatch, spaces):
    cfg(monkeypatch, 12, 4)
    L = 18
    H = models.long_range(L)
    arrs = marshal(H)
    left, right = Parity(spaces[0], L=L), Parity(spaces[1], L=L)
    x = rand_state(1 << (L - 1), seed=5)
    ref = orc.matvec(orc_msc(H), orc_sub(left), orc_sub(right), x, nthreads=4)
    mat = shell(H, left, right)
    assert "tiled=1" in mat.describe()
    y = mult_numpy(mat, x)
    assert np.max(np.abs(y - ref)) <= tol_for(arrs, x)


def test_spinconserve_generic_vs_oracle():
    L, k = 20, 10
    H = models.mbl(L)
    arrs = marshal(H)
    sub = SpinConserve(L, k)
    x = rand_state(sub.get_dimension(), seed=9)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    mat = shell(H, sub)
    y = mult_numpy(mat, x)
    assert np.max(np.abs(y - ref)) <= tol_for(arrs, x)
    assert abs(mat.norm() - orc.infnorm(orc_msc(H), orc_sub(sub), orc_sub(sub))) < 1e-12


def test_spinconserve_kernel_general_masks():
    """Sz-conserving operator with non-adjacent two-bit masks and four-bit masks:
    exercises the span loop of the incremental colex rank."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum
    L, k = 16, 7
    rs = np.random.RandomState(3)
    hop = lambda i, j: sigmax(i) * sigmax(j) + sigmay(i) * sigmay(j)
    H = op_sum(float(rs.uniform(-1, 1)) * hop(i, j) for i in range(L) for j in range(i + 1, L) if (i + j) % 3 != 0)
    H += op_sum(0.3 * hop(i, i + 2) * hop(i + 5, i + 9) for i in range(0, 6))
    H += op_sum(float(rs.uniform(-1, 1)) * sigmaz(i) * sigmaz((i + 4) % L) for i in range(L))
    H.L = L
    arrs = marshal(H)
    sub = SpinConserve(L, k)
    x = rand_state(sub.get_dimension(), seed=2)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    for diag in (False, True):
        mat = shell(H, sub)
        assert "SpinConserve kernel" in mat.describe()
        if diag:
            mat.precompute_diagonal()
        y = mult_numpy(mat, x)
        assert np.max(np.abs(y - ref)) <= tol_for(arrs, x)
        mat.destroy()
    for kk in (0, 1, L - 1, L):       # edge sectors (dimension 1 or L)
        sub = SpinConserve(L, kk)
        x = rand_state(sub.get_dimension(), seed=kk)
        ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
        y = mult_numpy(shell(H, sub), x)
        assert np.max(np.abs(y - ref)) <= tol_for(arrs, x)


@pytest.mark.parametrize("order", [0, 3])
@pytest.mark.parametrize("lb,L,k", [(10, 12, 6), (10, 16, 7), (10, 17, 3), (10, 18, 14), (13, 20, 10), (13, 20, 9),
                                    (13, 22, 11)])
def test_spinconserve_block_kernel(monkeypatch, lb, L, k, order):
    """Block form of the SpinConserve kernel (one workgroup per high part): chain bonds inside the low
    part (LDS), inside the high part (block runs), the bond across the boundary and long-range /
    four-spin masks (per-row path), with and without the cached diagonal, against the oracle."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum
    monkeypatch.setenv("DNM_SC_BLOCK", str(lb))
    monkeypatch.setenv("DNM_SC_ORDER", str(order))      # ascending high parts / equal-size groups
    rs = np.random.RandomState(L * 31 + k)
    hop = lambda i, j: sigmax(i) * sigmax(j) + sigmay(i) * sigmay(j)
    Hs = [models.mbl(L)]
    if L <= 18:
        E = op_sum(float(rs.uniform(-1, 1)) * hop(i, (i + 3) % L) for i in range(L))
        E += op_sum(0.3 * hop(i, i + 2) * hop(i + 5, i + 9) for i in range(0, L - 9, 2))
        E += op_sum(float(rs.uniform(-1, 1)) * sigmaz(i) * hop((i + 1) % L, (i + 2) % L) for i in range(L))
        E.L = L
        Hs.append(models.mbl(L) + E)
    sub = SpinConserve(L, k)
    x = rand_state(sub.get_dimension(), seed=L + k)
    for H in Hs:
        arrs = marshal(H)
        ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=4)
        for diag in (False, True):
            mat = shell(H, sub)
            assert "block form (%d" % lb in mat.describe()
            if diag:
                mat.precompute_diagonal()
            y = mult_numpy(mat, x)
            assert np.max(np.abs(y - ref)) <= tol_for(arrs, x)
            mat.destroy()


@pytest.mark.parametrize("P", [2, 3, 5])
def test_spinconserve_block_kernel_windows(monkeypatch, P):
    """The block kernel on a partition: blocks cut by the ownership boundaries, x given as a column window."""
    monkeypatch.setenv("DNM_SC_BLOCK", "10")
    monkeypatch.setenv("DNM_SC_ORDER", str(P % 2 * 2))   # both block orders
    L, k = 17, 8
    H = models.mbl(L)
    arrs = marshal(H)
    sub = SpinConserve(L, k)
    dim = sub.get_dimension()
    x = rand_state(dim, seed=16)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=4)
    Lb = _lib.lib()
    for diag in (False, True):
        y = np.empty(dim, dtype=complex)
        for r in range(P):
            h = backend.create_mat(*arrs, sub._c(), sub._c(), flags=0, rank=r, nranks=P)
            mat = backend.ShellMat(h, sub._c(), sub._c(), P, r)
            assert "block form" in mat.describe()
            start, n = backend.split_ownership(dim, P, r)
            lo, hi = mat.column_window()
            if diag:
                mat.precompute_diagonal()
            # guard amplitudes around the window: anything read outside it would poison the result
            xw = vec_from(x[lo:hi + 1])
            yl = vec_from(np.full(n, np.nan + 0j))
            _lib.check(Lb.dnm_mat_mult_window(mat.handle, xw.ptr, lo, hi - lo + 1, yl.ptr, None))
            y[start:start + n] = yl.local_numpy()
            mat.destroy()
        assert np.max(np.abs(y - ref)) <= tol_for(arrs, x)


def test_partitioned_kernels_on_one_gpu(monkeypatch):
    """Rank-local and partner passes of a P-way partition, run rank by rank on
    this one GPU (exchange = slicing), must add up to the single-rank result."""
    cfg(monkeypatch, 12, 4)
    L, P = 18, 4
    H = models.mbl(L)
    arrs = marshal(H)
    sub = Full(L=L)
    x = rand_state(1 << L, seed=4)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=4)
    nloc = (1 << L) // P
    Lb = _lib.lib()
    y = np.empty(1 << L, dtype=complex)
    xls = [vec_from(x[q * nloc:(q + 1) * nloc], sub.vec_swizzle) for q in range(P)]
    for r in range(P):
        h = backend.create_mat(*arrs, sub._c(), sub._c(), flags=0, rank=r, nranks=P)
        mat = backend.ShellMat(h, sub._c(), sub._c(), P, r)
        xl = xls[r]
        yl = backend.Vec(nloc, swz=mat.swz_left)
        _lib.check(Lb.dnm_mat_mult_local(mat.handle, xl.ptr, yl.ptr, None))
        for i, (p, off, cnt) in enumerate(mat.recvs):
            xr = partner_slice(xls[p], off, cnt)
            _lib.check(Lb.dnm_mat_mult_remote(mat.handle, i, xr.ptr, yl.ptr, None))
        y[r * nloc:(r + 1) * nloc] = yl.local_numpy()
        mat.destroy()
    assert np.max(np.abs(y - ref)) <= tol_for(arrs, x)


@pytest.mark.parametrize("P", [2, 3, 5])
def test_spinconserve_partitioned_windows(P):
    """Partitioned SpinConserve multiply rank by rank on this one GPU: PETSc-style
    ownership (uneven for P = 3, 5), the device-computed column window, and the
    windowed kernel must reproduce the single-rank oracle result."""
    L, k = 18, 9
    H = models.mbl(L)
    arrs = marshal(H)
    sub = SpinConserve(L, k)
    dim = sub.get_dimension()
    x = rand_state(dim, seed=6)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
    Lb = _lib.lib()
    import ctypes as C
    windows, owned = [], []
    for diag in (False, True):
        y = np.empty(dim, dtype=complex)
        for r in range(P):
            h = backend.create_mat(*arrs, sub._c(), sub._c(), flags=0, rank=r, nranks=P)
            mat = backend.ShellMat(h, sub._c(), sub._c(), P, r)
            start, n = backend.split_ownership(dim, P, r)
            assert (mat.row0, mat.m_local) == (start, n)
            lo, hi = mat.column_window()
            assert 0 <= lo <= start and start + n - 1 <= hi < dim
            if not diag:
                windows.append((lo, hi)); owned.append((start, n))
            if diag:
                mat.precompute_diagonal()
            xw = vec_from(x[lo:hi + 1])
            yl = vec_from(np.zeros(n, dtype=complex))
            _lib.check(Lb.dnm_mat_mult_window(mat.handle, xw.ptr, lo, hi - lo + 1, yl.ptr, None))
            y[start:start + n] = yl.local_numpy()
            with pytest.raises(_lib.BackendError):     # a window that misses needed columns is refused
                if hi - lo + 1 > n:
                    _lib.check(Lb.dnm_mat_mult_window(mat.handle, xw.ptr, lo + 1, hi - lo, yl.ptr, None))
                else:
                    raise _lib.BackendError("window equals block")
            mat.destroy()
        assert np.max(np.abs(y - ref)) <= tol_for(arrs, x)
    # the exchange schedule is consistent: what q receives from r is what r sends to q
    for me in range(P):
        recvs, _ = backend.window_exchange_ops(owned, windows, me)
        for src, lo, hi in recvs:
            _, sends = backend.window_exchange_ops(owned, windows, src)
            assert (me, lo, hi) in sends


def _sample_rows_check(H, L, xv, yv, nsamp=64, seed=0):
    """Recompute sampled rows of y = Hx on the host from the MSC definition
    (msc_tools.py:63-80) using x entries fetched from the device."""
    import torch
    masks, offs, signs, coeffs = marshal(H)
    rs = np.random.RandomState(seed)
    # random rows plus rows on both sides of every power-of-two boundary (tile, XCD group, window and swizzle
    # field edges all sit on those), each with random low bits
    edge = []
    for b in range(1, L):
        lowbits = int(rs.randint(0, 1 << b))
        edge += [(1 << b) - 1, 1 << b, ((1 << L) - 1) ^ (1 << b), (int(rs.randint(0, 1 << (L - b))) << b) | lowbits,
                 (((1 << (L - b)) - 1) << b) | lowbits]
    rows = np.unique(np.concatenate([[0, (1 << L) - 1], edge, rs.randint(0, 1 << L, nsamp)])).astype(np.int64)
    worst = 0.0
    ycheck = yv.array[yv.positions(torch.from_numpy(rows).to(yv.array.device))].cpu().numpy()
    for i, r in enumerate(rows):
        cols = r ^ masks
        xs = xv.array[xv.positions(torch.from_numpy(cols).to(xv.array.device))].cpu().numpy()
        acc = 0j
        for m in range(len(masks)):
            c = 0j
            for t in range(offs[m], offs[m + 1]):
                c += (1 - 2 * (bin(int(cols[m] & signs[t])).count("1") & 1)) * coeffs[t]
            acc += c * xs[m]
        worst = max(worst, abs(acc - ycheck[i]))
    return worst


@pytest.mark.default_layout
@pytest.mark.parametrize("L", [26, 30])
def test_full_size_properties(monkeypatch, L):
    """BASELINE configs 2/3 (L=26 XXZ, L=30 random-field Heisenberg): sampled
    rows against the MSC definition, Hermiticity <a|Hb> = <Ha|b>, linearity,
    and agreement between two different plans."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 5 * 16 * (1 << L):
        pytest.skip("not enough HBM")
    H = models.xxz(L) if L == 26 else models.mbl(L)
    sub = Full(L=L)
    n = 1 << L
    sw = sub.vec_swizzle
    a, b, Ha, Hb = (backend.Vec(n, swz=sw) for _ in range(4))
    a.set_random(1); b.set_random(2)
    na, nb = a.normalize(), b.normalize()
    cfg(monkeypatch, 12, 4, 0)
    mat = shell(H, sub)
    mat.mult(a, Ha)
    mat.mult(b, Hb)
    assert _sample_rows_check(H, L, a, Ha) < 1e-13
    lhs = Hb.dot(a)      # sum Hb_i conj(a_i) = <a|Hb>
    rhs = b.dot(Ha)      # <Ha|b>
    assert abs(lhs - rhs) < 1e-10
    # second plan (single pass, gathers) must agree element-wise
    cfg(monkeypatch, 13, 4, 1)
    mat2 = shell(H, sub)
    Ha2 = backend.Vec(n, swz=sw)
    mat2.mult(a, Ha2)
    Ha2.axpby(-1.0, 1.0, Ha)
    assert Ha2.norm() < 1e-12
    # linearity: H(a + 2i b) = Ha + 2i Hb
    a.axpby(2j, 1.0, b)
    mat.mult(a, Ha2)
    Ha.axpby(2j, 1.0, Hb)
    Ha2.axpby(-1.0, 1.0, Ha)
    assert Ha2.norm() < 1e-12
    mat.destroy(); mat2.destroy()


@pytest.mark.default_layout
@pytest.mark.parametrize("L", [26, 30])
def test_full_size_default_plan(monkeypatch, L):
    """The plan bench.py times (defaults: B=12, 4 rows per thread, LDS tiles + XCD-group gathers, fused
    dot product) against the plain multi-pass LDS plan, element-wise, at the BASELINE sizes."""
    import ctypes as C
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 4 * 16 * (1 << L):
        pytest.skip("not enough HBM")
    for k in ("DNM_TILE_BITS", "DNM_LOG_ROWS", "DNM_PLAN_MODE", "DNM_AMIN", "DNM_GBITS"):
        monkeypatch.delenv(k, raising=False)
    H = models.xxz(L) if L == 26 else models.mbl(L)
    sub = Full(L=L)
    n = 1 << L
    a, y0, y1 = (backend.Vec(n, swz=sub.vec_swizzle) for _ in range(3))
    a.set_random(4)
    a.normalize()
    mat = shell(H, sub)
    assert "mode=2" in mat.describe() and "B=12 logR=2" in mat.describe() and len(mat.describe().splitlines()) == 3
    d = (C.c_double * 2)()
    _lib.check(_lib.lib().dnm_mat_mult_dot(mat.handle, a.ptr, y0.ptr, d, None))
    ref_dot = a.dot(y0)                        # sum a_i conj(y0_i) = conj(<a, y0>)
    assert abs(complex(d[0], -d[1]) - ref_dot) < 1e-10
    cfg(monkeypatch, 12, 4, 0)
    mat2 = shell(H, sub)
    mat2.mult(a, y1)
    assert _sample_rows_check(H, L, a, y0) < 1e-13
    y1.axpby(-1.0, 1.0, y0)
    assert y1.norm() < 1e-12
    mat.destroy(); mat2.destroy()


@pytest.mark.default_layout
@pytest.mark.parametrize("L", [27, 30, 31])
def test_full_size_real_packed(monkeypatch, L):
    """The real-arithmetic operator at the sizes eigsolve uses it by default (L=30: 8 GiB vectors; L=31: two real
    amplitudes in each of 2^30 elements -- a size the complex form does not reach with a Krylov basis on one GPU):
    sampled rows against the MSC definition (row 2j + b is lane b of element j), symmetry of the real inner product,
    and the Lanczos step's fused sums."""
    import ctypes as C
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 4.5 * 8 * (1 << L):
        pytest.skip("not enough HBM")
    for k in PLAN_KNOBS:
        monkeypatch.delenv(k, raising=False)
    H = models.mbl(L)
    sub = Full(L=L)
    H.add_subspace(sub)
    mat = H.get_real_packed_mat(sub)
    assert mat is not None and mat.real_packed and mat.N == 1 << (L - 1)
    a, b = mat.createVecs()
    Ha, Hb = mat.createVecs()
    a.set_random(11); b.set_random(12)
    a.normalize(); b.normalize()
    mat.mult(a, Ha)
    mat.mult(b, Hb)
    masks, offs, signs, coeffs = marshal(H)
    assert np.abs(np.asarray(coeffs).imag).max() == 0.0
    rs = np.random.RandomState(L)
    edge = []
    for bit in range(1, L):
        edge += [(1 << bit) - 1, 1 << bit, ((1 << L) - 1) ^ (1 << bit), (int(rs.randint(0, 1 << (L - bit))) << bit) | int(rs.randint(0, 1 << bit))]
    rows = np.unique(np.concatenate([[0, 1, (1 << L) - 1], edge, rs.randint(0, 1 << L, 64)])).astype(np.int64)

    def lanes(v, idx):             # amplitudes idx of the packed vector v
        el = v.array[v.positions(torch.from_numpy(idx >> 1).to(v.array.device))].cpu().numpy()
        return np.where(idx & 1, el.imag, el.real)

    got = lanes(Ha, rows)
    worst = 0.0
    for i, r in enumerate(rows):
        cols = r ^ masks
        xs = lanes(a, cols)
        acc = 0.0
        for m in range(len(masks)):
            c = 0.0
            for t in range(offs[m], offs[m + 1]):
                c += (1 - 2 * (bin(int(cols[m] & signs[t])).count("1") & 1)) * coeffs[t].real
            acc += c * xs[m]
        worst = max(worst, abs(acc - got[i]))
    assert worst < 1e-13, worst
    # real symmetric: <a, H b> = <H a, b> (Vec.dot is the complex one: the real inner product is its real part when the
    # lanes are read as (re, im) of both factors -- sum_j re re + im im)
    assert abs(Hb.dot(a).real - b.dot(Ha).real) < 1e-10
    d = (C.c_double * 3)()
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, a.ptr, Hb.ptr, None, 0.0, d, None))
    want = Ha.dot(a).real
    assert abs(d[0] - want) < 1e-10 and abs(d[2] - Ha.dot(Ha).real) < 1e-9
    H.destroy_mat()


PLAN_KNOBS = ("DNM_TILE_BITS", "DNM_LOG_ROWS", "DNM_PLAN_MODE", "DNM_AMIN", "DNM_GBITS", "DNM_WINDOW_FIRST",
              "DNM_DIAG_PASS", "DNM_GBITS_WINDOW", "DNM_CACHE_POLICY", "DNM_SC_BLOCK", "DNM_SC_LAYOUT", "DNM_SWZ")

DEFAULT_PLAN_CASES = [(name, L) for L in (13, 16, 18, 20, 22, 24, 25) for name in ("mbl", "xxz", "heisenberg")] + \
                     [("long_range", 13), ("long_range", 16), ("long_range", 18), ("long_range", 20), ("syk", 12)]


@pytest.mark.default_layout
@pytest.mark.parametrize("name,L", DEFAULT_PLAN_CASES)
def test_default_plan_vs_oracle(monkeypatch, name, L):
    """The plan the production planner picks -- no DNM_* knob, production vector layout -- element-wise against the
    oracle at the sizes between the knob-driven tests (<= 2^20) and the full-size property tests (2^26, 2^30): the
    planner changes shape with the size (one pass up to 2^18; a second, window pass from 2^19; the window pass first
    from 2^25 on; group bits) and every one of those decisions is pinned here.  The reference tests every size the
    same way (tests/integration/test_multiply.py:85-106)."""
    for k in PLAN_KNOBS:
        monkeypatch.delenv(k, raising=False)
    H = models.BY_NAME[name](L)
    arrs = marshal(H)
    sub = Full(L=L)
    assert sub.vec_swizzle == 16
    n = 1 << L
    x = rand_state(n, seed=100 + L)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=min(16, orc.max_threads()))
    mat = shell(H, sub)
    d = mat.describe()
    lines = d.splitlines()
    if name == "syk":           # 2^12 amplitudes, 1820 four-Majorana strings: one tile, every mask inside it, as table records
        assert "tiled=1" in d and len(lines) == 3 and lines[2].startswith("table records: "), d
    else:
        assert "tiled=1" in d and "mode=2" in d and "B=12 logR=2" in d, d
    # where DESIGN.md section 4.1 says the plan changes shape
    if name not in ("syk", "long_range"):
        if L <= 18:
            assert len(lines) == 2 and "segs [0,12)" in lines[1], d
        elif L < 25:
            assert len(lines) == 3 and "segs [0,12)" in lines[1] and "segs [0,4) [%d,%d)" % (L - 8, L) in lines[2], d
            assert "acc=0" in lines[1] and "acc=1" in lines[2], d
        else:
            assert len(lines) == 3 and "segs [0,4) [17,25)" in lines[1] and "segs [0,12)" in lines[2], d
            assert "acc=0" in lines[1] and "acc=1" in lines[2] and "diag=1" in lines[2], d
    xv, yv = mat.createVecs()
    assert xv.swz == 16
    xv.set_local_from_numpy(x)
    yv.set(777.0)
    mat.mult(xv, yv)
    y = yv.local_numpy()
    assert np.max(np.abs(y - ref)) <= tol_for(arrs, x), d
    # the fused <x, y> of the same plan (what the Krylov loops call)
    import ctypes as C
    dd = (C.c_double * 2)()
    _lib.check(_lib.lib().dnm_mat_mult_dot(mat.handle, xv.ptr, yv.ptr, dd, None))
    want = np.vdot(x, ref)
    assert abs(complex(dd[0], dd[1]) - want) <= 1e-11 * max(1.0, abs(want)) * L
    assert np.max(np.abs(yv.local_numpy() - ref)) <= tol_for(arrs, x)
    mat.destroy()


@pytest.mark.default_layout
@pytest.mark.parametrize("L", [17, 21, 25])
def test_default_plan_parity_vs_oracle(monkeypatch, L):
    """Parity subspaces (an (L-1)-bit hypercube on the same tiled kernel) under the production planner."""
    for k in PLAN_KNOBS:
        monkeypatch.delenv(k, raising=False)
    H = models.ising(L)
    arrs = marshal(H)
    for space in (0, 1):
        sub = Parity(space, L=L)
        n = sub.get_dimension()
        x = rand_state(n, seed=200 + L + space)
        ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=min(16, orc.max_threads()))
        mat = shell(H, sub)
        assert "tiled=1" in mat.describe() and "n=%d" % (L - 1) in mat.describe()
        y = mult_numpy(mat, x)
        assert np.max(np.abs(y - ref)) <= tol_for(arrs, x), mat.describe()
        mat.destroy()


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_table_records_and_groups_gpu(monkeypatch, seed):
    """Random operators made to share masks (tests/fuzz_ops.py: table records of one group or several, 1-4 flipped spins, a
    long-range diagonal on top) under random tile shapes and plan modes, Full and Parity, against the oracle; the same with
    both record forms switched off."""
    from fuzz_ops import shared_mask_operator
    rs = np.random.RandomState(4200 + seed)
    L = int(rs.randint(13, 17))
    parity = seed % 2 == 1
    H = shared_mask_operator(rs, L, parity)
    sub = Parity(int(rs.randint(2)), L=L) if parity else Full(L=L)
    B, logR = [(8, 2), (10, 3), (10, 4), (12, 2), (12, 3), (11, 3)][int(rs.randint(6))]
    cfg(monkeypatch, B, logR)
    monkeypatch.setenv("DNM_PLAN_MODE", str(int(rs.choice([0, 1, 2]))))
    arrs = marshal(H)
    x = rand_state(sub.get_dimension(), seed=seed)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=4)
    seen_tabs = False
    for on in ("1", "0"):
        monkeypatch.setenv("DNM_TAB_RECORDS", on)
        monkeypatch.setenv("DNM_DIAG_GROUPS", on)
        mat = shell(H, sub)
        d = mat.describe()
        assert "tiled=1" in d and (on == "1" or "table records: " not in d), d
        seen_tabs = seen_tabs or "table records: " in d
        y = mult_numpy(mat, x)
        assert np.max(np.abs(y - ref)) <= tol_for(arrs, x) * L, d
        mat.destroy()
    assert seen_tabs, "no mask of this operator took the table form"


@pytest.mark.default_layout
@pytest.mark.parametrize("space", ["full", "even"])
def test_default_plan_grouped_diagonal_vs_oracle(monkeypatch, space):
    """Diagonal terms with one spin inside the tile and one outside (the harness's long_range: all-to-all ZZ), summed per
    group and workgroup (DevPass::gbucket) under the production planner, two-pass plan (2^19 / 2^20 amplitudes), against
    the oracle; DNM_DIAG_GROUPS=0 -- every term by every thread -- gives the same product."""
    for k in PLAN_KNOBS:
        monkeypatch.delenv(k, raising=False)
    L = 20
    H = models.long_range(L)
    arrs = marshal(H)
    sub = Full(L=L) if space == "full" else Parity(space, L=L)
    x = rand_state(sub.get_dimension(), seed=78)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=min(16, orc.max_threads()))
    ys = []
    for groups in ("1", "0"):
        monkeypatch.setenv("DNM_DIAG_GROUPS", groups)
        mat = shell(H, sub)
        assert "tiled=1" in mat.describe() and mat.describe().count("local pass") == 2, mat.describe()
        ys.append(mult_numpy(mat, x))
        assert np.max(np.abs(ys[-1] - ref)) <= tol_for(arrs, x) * L, mat.describe()
        mat.destroy()
    assert not np.array_equal(ys[0], ys[1])        # (two ways of summing the same terms: equal to rounding, not bit for bit)


@pytest.mark.default_layout
@pytest.mark.parametrize("space", ["full", "even", "odd"])
def test_default_plan_table_records_vs_oracle(monkeypatch, space):
    """Masks of many terms as table records (csrc/plan.h DevTab) under the production planner at a size whose plan gathers
    (SYK on 15 spins: 1 366 + 105 masks, tile of 12 bits, 8 rows per thread) -- the Full space and both Parity sectors (the
    dropped spin folded into the sign masks) element-wise against the oracle; the records of four terms give the same
    product."""
    for k in PLAN_KNOBS:
        monkeypatch.delenv(k, raising=False)
    L = 15
    H = models.syk(L)
    arrs = marshal(H)
    sub = Full(L=L) if space == "full" else Parity(space, L=L)
    x = rand_state(sub.get_dimension(), seed=77)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=min(16, orc.max_threads()))
    mat = shell(H, sub)
    d = mat.describe()
    assert "tiled=1" in d and "gather_masks=" in d and "gather_masks=0" not in d.splitlines()[1], d
    assert d.strip().splitlines()[-1].startswith("table records: "), d
    y = mult_numpy(mat, x)
    assert np.max(np.abs(y - ref)) <= tol_for(arrs, x), d
    mat.destroy()
    monkeypatch.setenv("DNM_TAB_RECORDS", "0")
    mat = shell(H, sub)
    assert "table records" not in mat.describe()
    y0 = mult_numpy(mat, x)
    assert np.max(np.abs(y0 - ref)) <= tol_for(arrs, x)
    mat.destroy()


@pytest.mark.default_layout
@pytest.mark.parametrize("L,k,internal", [(24, 12, False), (25, 12, True), (26, 13, True), (25, 9, False)])
def test_default_spinconserve_vs_oracle(monkeypatch, L, k, internal):
    """SpinConserve under the production configuration on both sides of the 2^22-state threshold of the internal
    three-field layout (config.sc_layout_min_dim): reference order and the row / block kernels below it, the (14, 10)
    layout and the two-pass kernels above -- element-wise against the oracle in reference order."""
    for kk in PLAN_KNOBS:
        monkeypatch.delenv(kk, raising=False)
    sub = SpinConserve(L, k)
    n = sub.get_dimension()
    assert (n >= 1 << 22) == internal
    assert (sub.vec_swizzle >= 256) == internal
    H = models.mbl(L)
    arrs = marshal(H)
    x = rand_state(n, seed=300 + L)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x, nthreads=min(16, orc.max_threads()))
    mat = shell(H, sub)
    xv, yv = mat.createVecs()
    assert xv.internal == internal
    xv.set_local_from_numpy(x)
    yv.set(777.0)
    mat.mult(xv, yv)
    assert np.max(np.abs(yv.local_numpy() - ref)) <= tol_for(arrs, x), (L, k)
    mat.destroy()


def test_spinconserve_large_properties():
    """SpinConserve L=28, k=14 (40 M amplitudes): the incremental-rank kernel against the generic
    row-gather kernel element-wise, with and without the cached diagonal, and Hermiticity."""
    L, k = 28, 14
    H = models.mbl(L)
    sub = SpinConserve(L, k)
    from gpu_util import vec_for
    a, b, Ha, Hb, Hg = (vec_for(sub) for _ in range(5))      # in the layout the subspace's states take
    a.set_random(1); b.set_random(2)
    a.normalize(); b.normalize()
    mat = shell(H, sub)
    matg = shell(H, sub, flags=_lib.MAT_FORCE_GATHER)        # works in reference order: vectors are converted
    assert "SpinConserve" in mat.describe() and "row-gather" in matg.describe()
    mat.mult(a, Ha)
    matg.mult(a, Hg)
    Hg.axpby(-1.0, 1.0, Ha)
    assert Hg.norm() < 1e-12
    mat.mult(b, Hb)
    assert abs(Hb.dot(a) - b.dot(Ha)) < 1e-10
    mat.precompute_diagonal()
    mat.mult(a, Hg)
    Hg.axpby(-1.0, 1.0, Ha)
    assert Hg.norm() < 1e-12
    mat.destroy(); matg.destroy()


@pytest.mark.default_layout
@pytest.mark.parametrize("name,L,kind", [("mbl", 20, "full"), ("heisenberg", 24, "full"), ("xxz", 25, "full"),
                                         ("heisenberg", 25, "parity"), ("mbl", 26, "parity")])
def test_default_plan_real_packed_vs_oracle(monkeypatch, name, L, kind):
    """The real-arithmetic operator eigsolve builds by default from 2^23 amplitudes on (Operator.get_real_packed_mat:
    no DNM_* knob, production layout of the packed vectors, the planner's own choice of passes for a vector of half
    the elements) element-wise against the oracle, with the Lanczos step's fused sums and the unpacking of a result
    into the complex state of the subspace's own layout."""
    import ctypes as C
    from dynamite_amd import backend
    for k in PLAN_KNOBS:
        monkeypatch.delenv(k, raising=False)
    H = models.BY_NAME[name](L)
    sub = Full(L=L) if kind == "full" else Parity('even', L=L)
    H.add_subspace(sub)
    arrs = marshal(H)
    dim = sub.get_dimension()
    mat = H.get_real_packed_mat(sub)
    assert mat is not None and mat.real_packed and mat.N == dim // 2
    d = mat.describe()
    assert "tiled=1" in d and "B=12 logR=2" in d, d
    rs = np.random.RandomState(L)
    xr = rs.standard_normal(dim)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xr.astype(np.complex128), nthreads=min(16, orc.max_threads())).real
    xv, yv = mat.createVecs()
    xv.set_local_from_numpy(xr[0::2] + 1j * xr[1::2])
    yv.set(777.0)
    mat.mult(xv, yv)
    yp = yv.local_numpy()
    tol = tol_for(arrs, xr)
    assert max(np.abs(yp.real - ref[0::2]).max(), np.abs(yp.imag - ref[1::2]).max()) <= tol, d
    dd = (C.c_double * 3)()
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, xv.ptr, yv.ptr, None, 0.0, dd, None))
    assert abs(dd[0] - xr @ ref) <= 1e-11 * L * max(1.0, abs(xr @ ref)) and abs(dd[2] - ref @ ref) <= 1e-11 * L * (ref @ ref)
    out = backend.Vec(dim, swz=sub.vec_swizzle)
    _lib.check(_lib.lib().dnm_vec_unpack_real(out.ptr, yv.ptr, mat.n_local, mat.swz_right, sub.vec_swizzle, None))
    got = out.local_numpy()
    assert np.abs(got.real - ref).max() <= tol and np.abs(got.imag).max() == 0.0
    H.destroy_mat()


def test_row_kernels_launched_in_slices(monkeypatch):
    """The cached diagonal and CheckConserves run one thread per row / column, and a launch holds fewer than 2^32 threads:
    past 2^30 rows they go out in slices (round 5: XParity(SpinConserve(36, 18)) has 4.54 G rows, and its first solve ran
    with no diagonal at all).  Slices of 2^10 here: the same results as in one launch."""
    monkeypatch.setenv("DNM_LAUNCH_SLICE_LOG2", "10")
    for sub in (SpinConserve(16, 8), Full(L=13), Explicit(np.arange(0, 1 << 14, 3), L=14)):
        H = models.mbl(sub.L)
        mat = shell(H, sub, flags=_lib.MAT_FORCE_GATHER)
        mat.precompute_diagonal()
        d = np.empty(sub.get_dimension())
        _lib.check(_lib.lib().dnm_mat_get_diagonal(mat.handle, d.ctypes.data_as(_lib.f64p), None))
        want = orc.precompute_diagonal(orc_msc(H), orc_sub(sub))
        assert np.abs(d - want).max() <= 1e-13
        x = rand_state(sub.get_dimension(), seed=2)
        assert np.abs(mult_numpy(mat, x) - orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)).max() <= 1e-12
        mat.destroy()
        assert backend.check_conserves(*marshal(H), sub._to_c(), sub._to_c()) == orc.check_conserves(
            orc_msc(H), orc_sub(sub), orc_sub(sub))
    H = models.ising(12)            # sigma_x leaves SpinConserve: the verdict must survive the slicing too
    sub = SpinConserve(12, 6)
    assert backend.check_conserves(*marshal(H), sub._to_c(), sub._to_c()) is False


def _random_hermitian(L, nterms, rs):
    """Sum of random Pauli strings with real coefficients (Hermitian by construction)."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum, op_product
    terms = []
    for _ in range(nterms):
        w = rs.randint(1, min(L, 5) + 1)
        sites = rs.choice(L, size=w, replace=False)
        ops = [(sigmax, sigmay, sigmaz)[rs.randint(3)](int(i)) for i in sites]
        terms.append(float(rs.uniform(-1, 1)) * op_product(ops))
    H = op_sum(terms)
    H.L = L
    return H


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_FUZZ_N", "64"))))
def test_fuzz_random_operators(monkeypatch, seed):
    """Random Pauli-string Hamiltonians (long strings, imaginary matrix elements, many terms per mask,
    masks straddling tile / group / window boundaries) on random subspaces, tile shapes and plan modes,
    rank-partitioned or not, against the CPU oracle."""
    rs = np.random.RandomState(100 + seed)
    L = int(rs.randint(9, 19))
    B, logR = [(8, 2), (10, 2), (10, 3), (11, 3), (12, 3), (12, 4)][rs.randint(6)]
    mode = int(rs.randint(3))
    cfg(monkeypatch, B=B, logR=logR, mode=mode, amin=int(rs.randint(3, 6)), gbits=int(rs.randint(0, 7)))
    monkeypatch.setenv("DNM_CACHE_POLICY", str([0, 32, 226][rs.randint(3)]))
    H = _random_hermitian(L, int(rs.randint(3, 40)), rs)
    arrs = marshal(H)
    kind = ["full", "full", "parity", "sc", "explicit"][rs.randint(5)]
    if kind == "full":
        left = right = Full(L=L)
    elif kind == "parity":
        left, right = Parity(int(rs.randint(2)), L=L), Parity(int(rs.randint(2)), L=L)
    elif kind == "sc":
        left = right = SpinConserve(L, int(rs.randint(1, L)))
    else:
        left = right = Explicit(np.sort(rs.choice(1 << L, size=min(1 << L, 3000), replace=False)), L=L)
    x = rand_state(right.get_dimension(), seed=seed)
    ref = orc.matvec(orc_msc(H), orc_sub(left), orc_sub(right), x)
    mat = shell(H, left, right)
    y = mult_numpy(mat, x)
    assert np.max(np.abs(y - ref)) <= tol_for(arrs, x), (L, B, logR, mode, kind, mat.describe())
    assert abs(mat.norm() - orc.infnorm(orc_msc(H), orc_sub(left), orc_sub(right))) <= 1e-12 * max(1.0, mat.norm())
    mat.destroy()
    # the same operator partitioned over P ranks (hypercube subspaces with a local block >= one tile)
    P = int(2 ** rs.randint(1, 3))
    n = L if kind == "full" else L - 1
    if kind in ("full", "parity") and left is right or (kind == "parity" and left.space == right.space):
        if n - int(np.log2(P)) - 1 >= B:
            Lb = _lib.lib()
            nloc = left.get_dimension() // P
            yp = np.empty(left.get_dimension(), dtype=complex)
            xls = [vec_from(x[q * nloc:(q + 1) * nloc], right.vec_swizzle) for q in range(P)]
            for r in range(P):
                h = backend.create_mat(*arrs, left._c(), right._c(), flags=0, rank=r, nranks=P)
                m = backend.ShellMat(h, left._c(), right._c(), P, r)
                xl, yl = xls[r], backend.Vec(nloc, swz=m.swz_left)
                _lib.check(Lb.dnm_mat_mult_local(m.handle, xl.ptr, yl.ptr, None))
                for i, (p, off, cnt) in enumerate(m.recvs):
                    xr = partner_slice(xls[p], off, cnt)
                    _lib.check(Lb.dnm_mat_mult_remote(m.handle, i, xr.ptr, yl.ptr, None))
                yp[r * nloc:(r + 1) * nloc] = yl.local_numpy()
                m.destroy()
            assert np.max(np.abs(yp - ref)) <= tol_for(arrs, x), (L, B, P, kind)
    if kind == "sc":
        # SpinConserve partitioned by column windows (uneven PETSc-style ownership)
        Pw = int(rs.randint(2, 5))
        dim = left.get_dimension()
        if dim >= 4 * Pw:
            Lb = _lib.lib()
            yp = np.empty(dim, dtype=complex)
            for r in range(Pw):
                h = backend.create_mat(*arrs, left._c(), right._c(), flags=0, rank=r, nranks=Pw)
                m = backend.ShellMat(h, left._c(), right._c(), Pw, r)
                start, n = backend.split_ownership(dim, Pw, r)
                lo, hi = m.column_window()
                assert 0 <= lo <= start and start + n - 1 <= hi < dim
                xw, yl = vec_from(x[lo:hi + 1]), vec_from(np.zeros(n, dtype=complex))
                _lib.check(Lb.dnm_mat_mult_window(m.handle, xw.ptr, lo, hi - lo + 1, yl.ptr, None))
                yp[start:start + n] = yl.local_numpy()
                m.destroy()
            assert np.max(np.abs(yp - ref)) <= tol_for(arrs, x), (L, Pw, "sc windows")


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_FUZZ_XPARITY_N", "24"))))
def test_fuzz_xparity_random_operators(monkeypatch, seed):
    """Random Pauli-string Hamiltonians inside XParity sectors of Full, Parity and half-filled SpinConserve parents
    (subspaces.py:632-674: strings that anticommute with the global flip are projected away, strings that flip spin L-1
    come composed with the flip -- masks of up to L-1 bits): Operator.dot against the oracle's xparity(parent) on the
    reduced operator (whose construction tests/golden/xparity.npz pins), on random tile shapes and plan modes; for real
    symmetric operators the lowest level in real arithmetic (DNM_MAT_REAL_PACKED under XParity) against a dense solve."""
    from dynamite_amd import msc_tools
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import XParity
    from dynamite_amd.computations import eigsolve
    rs = np.random.RandomState(700 + seed)
    L = 2 * int(rs.randint(5, 7))                      # 10, 12 (Parity and SpinConserve parents need an even L; the dense
    #                                                    solve on the host sets the size)
    B, logR = [(8, 2), (10, 2), (10, 3)][rs.randint(3)]
    cfg(monkeypatch, B=B, logR=logR, mode=int(rs.randint(3)), amin=int(rs.randint(3, 5)), gbits=int(rs.randint(0, 4)))
    real = seed % 2 == 0
    H = (_random_real_symmetric if real else _random_hermitian)(L, int(rs.randint(6, 40)), rs)
    kind = ["full", "parity", "sc"][rs.randint(3)]
    parent = {"full": Full(L=L), "parity": Parity(int(rs.randint(2)), L=L), "sc": SpinConserve(L, L // 2)}[kind]
    sub = XParity(parent, sector=['+', '-'][rs.randint(2)])
    H.establish_L()
    H.reduce_msc()
    m = sub.reduce_msc(H.msc)
    if m.size == 0:
        return                                          # every string anticommuted with the flip
    H.add_subspace(sub)
    H.allow_projection = True
    x = rand_state(sub.get_dimension(), seed=seed)
    xs = State(L=L, subspace=sub)
    xs.vec.set_local_from_numpy(x)
    xs.set_initialized()
    y = H.dot(xs).to_numpy()
    masks, offs = msc_tools.get_mask_offsets(m)
    osub = orc.xparity(orc_sub(parent))
    red = orc.Msc(masks, offs, m['signs'], m['coeffs'])
    ref = orc.matvec(red, osub, osub, x)
    assert np.max(np.abs(y - ref)) <= tol_for((masks, offs, m['signs'], m['coeffs']), x), (L, kind, B, logR, H.get_mat(subspaces=(sub, sub)).describe())
    if real and kind != "sc":
        A = H.to_numpy(subspaces=(sub, sub), sparse=False)
        if np.abs(A - A.conj().T).max() < 1e-12 and np.abs(A.imag).max() == 0 and np.ptp(np.linalg.eigvalsh(A)) > 1e-6:
            monkeypatch.setenv("DNM_EIGS_REAL", "1")
            ev = H.eigsolve(nev=1, tol=1e-11, subspace=sub)
            w = np.linalg.eigvalsh(A)
            assert abs(ev[0] - w[0]) < 1e-9 * max(1.0, abs(w[0])), (L, kind)
            if eigsolve.last_stats['real_arithmetic'] is not True:
                # (a projection that drops strings can leave an operator the packed form refuses, and a packed
                # vector of L - 2 index bits can be too short for the tile: complex ran)
                assert kind == "parity" or L - 2 <= B, (L, B, kind)
    H.destroy_mat()


def _random_real_symmetric(L, nterms, rs):
    """Sum of random Pauli strings with an EVEN number of sigma_y each and real coefficients: a real symmetric matrix
    (many terms per mask, signs that reach bit 0, masks that flip bit 0 alone, diagonal strings)."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum, op_product
    terms = []
    for _ in range(nterms):
        w = rs.randint(1, min(L, 5) + 1)
        sites = [int(i) for i in rs.choice(L, size=w, replace=False)]
        if rs.randint(4) == 0:
            sites[0] = 0                                     # bit 0 is the packed bit: make it busy
            sites = list(dict.fromkeys(sites))
        kinds = [int(rs.randint(3)) for _ in sites]
        if sum(k == 1 for k in kinds) % 2:
            j = kinds.index(1)
            kinds[j] = 0 if rs.randint(2) else 2
        terms.append(float(rs.uniform(-1, 1)) * op_product([(sigmax, sigmay, sigmaz)[k](i) for k, i in zip(kinds, sites)]))
    H = op_sum(terms)
    H.L = L
    return H


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DNM_FUZZ_REAL_N", "32"))))
def test_fuzz_real_packed_operators(monkeypatch, seed):
    """DNM_MAT_REAL_PACKED (the real-arithmetic form eigsolve uses, two real amplitudes per element) on random real
    symmetric Pauli sums, random tile shapes and plan modes, Full and Parity, whole and partitioned over 2 / 4 ranks
    (partner blocks), against the oracle's complex multiply of the same real vector."""
    rs = np.random.RandomState(4000 + seed)
    B, logR = [(8, 2), (10, 2), (10, 3), (11, 3), (12, 3), (12, 4)][rs.randint(6)]
    L = int(rs.randint(B + 2, 19))            # the packed form runs on the tiled kernel only: >= B packed index bits
    mode = int(rs.randint(3))
    cfg(monkeypatch, B=B, logR=logR, mode=mode, amin=int(rs.randint(3, 6)), gbits=int(rs.randint(0, 7)))
    monkeypatch.setenv("DNM_CACHE_POLICY", str([0, 32, 226][rs.randint(3)]))
    H = _random_real_symmetric(L, int(rs.randint(3, 40)), rs)
    arrs = marshal(H)
    sub = Full(L=L) if rs.randint(2) else Parity(int(rs.randint(2)), L=L)
    dim = sub.get_dimension()
    xr = rs.standard_normal(dim)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xr.astype(np.complex128))
    assert np.abs(ref.imag).max() == 0.0
    ref = ref.real
    tol = tol_for(arrs, xr)

    def unpack(yp):
        got = np.empty(2 * yp.shape[0])
        got[0::2], got[1::2] = yp.real, yp.imag
        return got

    xp = xr[0::2] + 1j * xr[1::2]
    mat = shell(H, sub, flags=_lib.MAT_REAL_PACKED)
    assert mat.real_packed and mat.N == dim // 2
    y = unpack(mult_numpy(mat, xp))
    assert np.abs(y - ref).max() <= tol, (L, B, logR, mode, mat.describe())
    mat.destroy()
    P = int(2 ** rs.randint(1, 3))
    n = (L if isinstance(sub, Full) else L - 1) - 1              # index bits of the packed operator
    if n - int(np.log2(P)) - 1 >= B:
        Lb = _lib.lib()
        nloc = dim // 2 // P
        yp = np.empty(dim // 2, dtype=complex)
        c = sub._c()
        xls = None
        for r in range(P):
            h = backend.create_mat(*arrs, c, c, flags=_lib.MAT_REAL_PACKED, rank=r, nranks=P)
            m = backend.ShellMat(h, c, c, P, r)
            assert m.n_local == nloc
            if xls is None:
                xls = [vec_from(xp[q * nloc:(q + 1) * nloc], m.swz_right) for q in range(P)]
            yl = backend.Vec(nloc, swz=m.swz_left)
            _lib.check(Lb.dnm_mat_mult_local(m.handle, xls[r].ptr, yl.ptr, None))
            for i, (p_, off, cnt) in enumerate(m.recvs):
                xq = partner_slice(xls[p_], off, cnt)
                _lib.check(Lb.dnm_mat_mult_remote(m.handle, i, xq.ptr, yl.ptr, None))
            yp[r * nloc:(r + 1) * nloc] = yl.local_numpy()
            m.destroy()
        assert np.abs(unpack(yp) - ref).max() <= tol, (L, B, P, "partitioned")


def test_error_behaviour():
    H = models.mbl(12)
    sub = Full(L=12)
    mat = shell(H, sub)
    x = vec_from(rand_state(1 << 12), mat.swz_right)
    with pytest.raises(ValueError):
        mat.mult(x, x)
    with pytest.raises(ValueError):
        mat.norm('frobenius')
    mat.destroy()
    with pytest.raises(RuntimeError):
        mat.mult(x, backend.Vec(1 << 12, swz=x.swz))
    # unsorted masks are rejected by the native layer
    arrs = marshal(H)
    bad = (arrs[0][::-1].copy(),) + arrs[1:]
    with pytest.raises(_lib.BackendError):
        backend.create_mat(*bad, sub._c(), sub._c())


@pytest.mark.parametrize("name,L,sub", [("mbl", 16, "full"), ("long_range", 12, "full"), ("ising", 14, "parity"),
                                        ("mbl", 6, "full"), ("heisenberg", 12, "sc"), ("mbl", 15, "scblock"),
                                        ("heisenberg", 13, "scblock")])
def test_mult_dot_fused(monkeypatch, name, L, sub):
    """dnm_mat_mult_dot: y identical to dnm_mat_mult, <x, y> equal to the separate dot product
    (fused into the last tiled pass when x is staged there; separate sweep otherwise)."""
    import ctypes as C
    cfg(monkeypatch, B=8, logR=2, mode=2, amin=3, gbits=3)
    H = models.BY_NAME[name](L)
    if sub == "scblock":        # the block form of the SpinConserve kernel takes the sums while x is in LDS
        monkeypatch.setenv("DNM_SC_BLOCK", "10")
        monkeypatch.setenv("DNM_SC_ORDER", "2")
    s = {"full": Full(L=L), "parity": Parity('even', L=L), "sc": SpinConserve(L, L // 2),
         "scblock": SpinConserve(L, L // 2)}[sub]
    mat = shell(H, s)
    if sub == "scblock":
        assert "block form" in mat.describe()
        mat.precompute_diagonal()
    x = rand_state(s.get_dimension(), seed=8)
    xv, y1, y2 = vec_from(x, mat.swz_right), backend.Vec(mat.M, swz=mat.swz_left), backend.Vec(mat.M, swz=mat.swz_left)
    mat.mult(xv, y1)
    d = (C.c_double * 2)()
    _lib.check(_lib.lib().dnm_mat_mult_dot(mat.handle, xv.ptr, y2.ptr, d, None))
    assert np.array_equal(y1.local_numpy(), y2.local_numpy())
    ref = np.vdot(x, y1.local_numpy())
    assert abs(complex(d[0], d[1]) - ref) <= 1e-13 * max(1.0, abs(ref)) * np.sqrt(x.size)
    assert abs(d[1]) <= 1e-12 * max(1.0, abs(ref))      # Hermitian operator: real expectation value
    # the whole Lanczos multiply: y = Hx - b z, <x, y>
    z = rand_state(s.get_dimension(), seed=9)
    zv, y3 = vec_from(z, mat.swz_left), backend.Vec(mat.M, swz=mat.swz_left)
    d3 = (C.c_double * 3)()
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, xv.ptr, y3.ptr, zv.ptr, 0.37, d3, None))
    want = y1.local_numpy() - 0.37 * z
    assert np.max(np.abs(y3.local_numpy() - want)) <= 4e-16 * max(1.0, np.abs(want).max())
    ref = np.vdot(x, want)
    assert abs(complex(d3[0], d3[1]) - ref) <= 1e-13 * max(1.0, abs(ref)) * np.sqrt(x.size)
    assert abs(d3[2] - np.vdot(want, want).real) <= 1e-13 * np.vdot(want, want).real      # |y|^2 rides along
    _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, xv.ptr, y3.ptr, None, 0.0, d3, None))
    assert np.array_equal(y1.local_numpy(), y3.local_numpy())
    nn = np.vdot(y1.local_numpy(), y1.local_numpy()).real
    assert abs(d3[2] - nn) <= 1e-13 * nn and abs(complex(d3[0], d3[1]) - np.vdot(x, y1.local_numpy())) <= 1e-12 * nn
    mat.destroy()
