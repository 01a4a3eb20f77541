"""
CPU tests of the product's host side: native subspace maps (bit-exact against
the reference's tables and the oracle), operator marshalling (identical arrays
to the reference's, from tests/golden), and the planner + pass tables, executed
by a numpy emulation of the tiled kernel and compared with the oracle.
"""
import numpy as np
import pytest

from dynamite_amd import _lib, models, msc_tools
from dynamite_amd.subspaces import Full, Parity, SpinConserve, Explicit
from oracle import oracle as orc
from plan_emulator import HostMat, multiply, run_pass, run_remote

EPS = 2.2e-16


# ------------------------------------------------------------------ subspaces (bit-exact)

def test_parity_tables(known):
    k = known["parity"]
    for space in (0, 1):
        sp = Parity(space, L=k["L"])
        correct = np.array([int(s, 2) for s in k["states"][str(space)]], dtype=np.int64)
        assert sp.get_dimension() == len(correct)
        assert np.array_equal(sp.idx_to_state(np.arange(len(correct))), correct)
        assert np.array_equal(sp.state_to_idx(correct), np.arange(len(correct)))
        assert sp.state_to_idx(int(k["bad_states"][str(space)], 2)) == -1


def test_spin_conserve_tables(known):
    k = known["spin_conserve"]
    for L, kk, dim in k["dims"]:
        assert SpinConserve(L, kk).get_dimension() == dim
    s = k["single"]
    sp = SpinConserve(s["L"], s["k"])
    assert sp.idx_to_state(s["idx"]) == int(s["state"], 2)
    assert sp.state_to_idx(int(s["state"], 2)) == s["idx"]
    for c in k["invalid_s2i"]:
        assert SpinConserve(c["L"], c["k"]).state_to_idx(int(c["state"], 2)) == -1
    for kk in (1, 2):
        correct = np.array([int(x, 2) for x in k["tables"][str(kk)]], dtype=np.int64)
        sp = SpinConserve(k["tables"]["L"], kk)
        assert np.array_equal(sp.idx_to_state(np.arange(len(correct))), correct)
        assert np.array_equal(sp.state_to_idx(correct), np.arange(len(correct)))
        for idx in (-1, sp.get_dimension()):
            with pytest.raises(ValueError):
                sp.idx_to_state(idx)


def test_explicit_tables(known):
    k = known["explicit"]
    uns = [int(s, 2) for s in k["unsorted"]]
    for states in (uns, sorted(uns)):
        sp = Explicit(states, L=k["L"])
        assert sp.get_dimension() == len(states)
        assert np.array_equal(sp.state_to_idx(states), np.arange(len(states)))
        assert np.array_equal(sp.idx_to_state(np.arange(len(states))), np.array(states))
        assert sp.state_to_idx(int(k["bad_state"], 2)) == -1
    with pytest.raises(ValueError):
        Explicit([0, 1, 2, 2])
    with pytest.raises(ValueError):
        Explicit(uns, L=4)


def test_maps_equal_oracle_bit_exact():
    rs = np.random.RandomState(3)
    cases = [(SpinConserve(20, 9), orc.spin_conserve(20, 9)),
             (SpinConserve(36, 18), orc.spin_conserve(36, 18)),
             (Parity(1, L=30), orc.parity(30, 1)),
             (Full(L=33), orc.full(33))]
    for mine, ref in cases:
        dim = mine.get_dimension()
        assert dim == ref.dim
        idx = np.unique(np.concatenate([[0, dim - 1], rs.randint(0, min(dim, 2 ** 62), 2000) % dim]))
        st = mine.idx_to_state(idx)
        assert np.array_equal(st, ref.i2s(idx))
        assert np.array_equal(mine.state_to_idx(st), idx)
        probe = rs.randint(0, 2 ** 62, 2000) % (1 << mine.L)
        assert np.array_equal(mine.state_to_idx(probe), ref.s2i(probe))
    states = np.sort(rs.choice(1 << 16, 500, replace=False))
    perm = rs.permutation(states)
    for st in (states, perm):
        mine, ref = Explicit(st, L=16), orc.explicit(16, st)
        probe = rs.randint(0, 1 << 16, 3000)
        assert np.array_equal(mine.state_to_idx(probe), ref.s2i(probe))


# ------------------------------------------------------------------ marshalling = reference arrays

CASES = [('mbl', 6), ('mbl', 10), ('mbl', 12), ('heisenberg', 10), ('xxz', 10), ('ising', 10),
         ('long_range', 8), ('localized', 10), ('syk', 5), ('xsum', 8)]


@pytest.mark.parametrize("name,L", CASES)
def test_marshalled_arrays_identical_to_reference(golden_full, name, L):
    g = golden_full[f"{name}_L{L}"]
    H = models.BY_NAME[name](L)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    assert np.array_equal(masks, g["masks"])
    assert np.array_equal(offs, g["mask_offsets"])
    assert np.array_equal(H.msc["signs"], g["signs"])
    assert np.array_equal(H.msc["coeffs"], g["coeffs"])        # bit-identical doubles
    assert np.array_equal(np.frombuffer(H.serialize(), dtype=np.uint8), g["serialized"])
    assert msc_tools.is_hermitian(H.msc) == bool(g["hermitian"])
    back = msc_tools.deserialize(H.serialize())
    assert np.array_equal(back, H.msc)


def test_to_numpy_matches_golden(golden_full):
    g = golden_full["mbl_L10"]
    H = models.mbl(10)
    A = H.to_numpy()
    assert np.max(np.abs(A @ g["x"] - g["y"])) < 1e-13


def test_algebra_pauli(known):
    from dynamite_amd.operators import sigmax, sigmay, sigmaz
    X, Y, Z = sigmax(), sigmay(), sigmaz()
    for a, b, c in ((X, Y, Z), (Y, Z, X), (Z, X, Y)):
        assert (a * b) == 1j * c
        assert (a * a) == 1 * (X * X)
    for name, op in (("sigmax", X), ("sigmay", Y), ("sigmaz", Z)):
        op.L = 1
        from conftest import cmatrix
        assert np.array_equal(op.to_numpy(sparse=False), cmatrix(known["pauli"][name]["matrix"]))


# ------------------------------------------------------------------ planner + tables via emulator

def _orc_msc(H):
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    return orc.Msc(masks, offs, H.msc["signs"], H.msc["coeffs"]), (masks, offs, H.msc["signs"], H.msc["coeffs"])


def _rand(n, seed=0):
    rs = np.random.RandomState(seed)
    return rs.standard_normal(n) + 1j * rs.standard_normal(n)


def _cfg(monkeypatch, B, logR, mode=0, amin=3):
    monkeypatch.setenv("DNM_TILE_BITS", str(B))
    monkeypatch.setenv("DNM_LOG_ROWS", str(logR))
    monkeypatch.setenv("DNM_PLAN_MODE", str(mode))
    monkeypatch.setenv("DNM_AMIN", str(amin))


@pytest.mark.parametrize("B,logR,mode", [(8, 2, 0), (10, 3, 0), (10, 4, 0), (12, 4, 0), (8, 2, 1), (10, 3, 1),
                                         (8, 2, 2), (10, 3, 2)])
@pytest.mark.parametrize("name", ["mbl", "long_range", "ising", "syk"])
def test_tiled_plan_full_space(monkeypatch, name, B, logR, mode):
    L = 13 if name != "syk" else 12
    _cfg(monkeypatch, B, logR, mode)
    H = models.BY_NAME[name](L if name != "syk" else 6)
    if name == "syk":
        from dynamite_amd.operators import Operator
        H = Operator(msc=H.msc)
        H.L = L       # pad: identity on the upper spins
    omsc, arrs = _orc_msc(H)
    sub = Full(L=L)
    hm = HostMat(*arrs, sub._c(), sub._c())
    assert hm.tiled == 1, hm.describe()
    x = _rand(1 << L)
    y = multiply(hm, x)
    osub = orc.full(L)
    ref = orc.matvec_general(omsc, osub, osub, x)
    tol = 64 * len(arrs[0]) * EPS * np.abs(arrs[3]).max() * np.abs(x).max()
    assert np.max(np.abs(y - ref)) <= tol, hm.describe()


@pytest.mark.parametrize("B,logR,mode,P", [(8, 2, 2, 1), (8, 2, 1, 1), (8, 2, 0, 1), (10, 3, 2, 1), (10, 4, 1, 1),
                                           (8, 2, 2, 2), (8, 2, 2, 4), (8, 2, 1, 2)])
def test_table_records_syk(monkeypatch, B, logR, mode, P):
    """Masks of many terms as table records (plan.h: DevTab; SYK: sixteen Majorana products per set of four flipped
    spins): the emulation of the kernel's table arithmetic on the exported records against the oracle, with flipped bits in
    the tile, in the block part and -- partitioned -- among the rank bits; DNM_TAB_RECORDS=0 gives the records of four
    terms and the same product."""
    L = 11
    _cfg(monkeypatch, B, logR, mode)
    H = models.syk(L)
    omsc, arrs = _orc_msc(H)
    sub = Full(L=L)
    x = _rand(1 << L, 3)
    ref = orc.matvec_general(omsc, orc.full(L), orc.full(L), x)
    tol = 64 * len(arrs[0]) * EPS * np.abs(arrs[3]).max() * np.abs(x).max()
    nloc = (1 << L) // P
    kinds = set()
    for use_tabs in (True, False):
        monkeypatch.setenv("DNM_TAB_RECORDS", "1" if use_tabs else "0")
        y = np.zeros(1 << L, dtype=complex)
        nrec = 0
        for r in range(P):
            hm = HostMat(*arrs, sub._c(), sub._c(), rank=r, nranks=P)
            assert hm.tiled == 1, hm.describe()
            yl = np.zeros(nloc, dtype=complex)
            for p in hm.local:
                run_pass(hm, p, x[r * nloc:(r + 1) * nloc], yl)
            for i, (partner, off, cnt) in enumerate(hm.recvs):
                run_remote(hm, i, x[partner * nloc + off:partner * nloc + off + cnt], yl)
            y[r * nloc:(r + 1) * nloc] = yl
            for desc, quads in hm.local + hm.remote:
                nrec += desc.loop[_lib.LP_COUNT] - desc.loop[0]
                assert (desc.tab_loop[2] > 0) == (len(quads.tabs) > 0)
                if not use_tabs:
                    assert desc.tab_loop[2] == 0
                for q, T in enumerate(quads.tabs):
                    kinds.add(("gather" if q >= desc.tab_loop[1] else "tile", T.nbits, "ext" if T.ewid else "in",
                               "k" if T.flags & 1 else "t"))
        assert np.max(np.abs(y - ref)) <= tol
        if use_tabs:
            with_tabs = nrec
        else:
            assert nrec > 2 * with_tabs         # the records of four terms the tables replaced
    # every kind of table record was met: partners from the tile and gathered, 2 and 4 flipped bits, table bits in the
    # thread part, among a thread's rows (k bits) and outside the tile
    assert {k[0] for k in kinds} == {"tile", "gather"} and {k[1] for k in kinds} == {2, 4}
    assert {k[2] for k in kinds} == {"ext", "in"} and {k[3] for k in kinds} == {"k", "t"}


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_table_records_and_groups(monkeypatch, seed):
    """Random operators made to share masks: for a handful of sets of 1-4 flipped spins, 5-24 Pauli strings each (X or Y at
    every flipped spin, Z strings from a small pool elsewhere -- so that table records have one group or several, of one term
    or many), a random long-range diagonal on top (grouped diagonal terms), random tile shapes / plan modes / rank counts, Full
    and Parity: the emulation of the exported records against the oracle, and the same with both forms switched off."""
    from fuzz_ops import shared_mask_operator
    rs = np.random.RandomState(4200 + seed)
    L = int(rs.randint(10, 13))
    parity = seed % 2 == 1                  # (a Parity sector needs every mask to flip an even number of spins)
    H = shared_mask_operator(rs, L, parity)
    sub = Parity(int(rs.randint(2)), L=L) if parity else Full(L=L)
    osub = orc.parity(L, sub.space) if parity else orc.full(L)
    n = L - 1 if parity else L
    B, logR = [(8, 2), (10, 3), (10, 4), (8, 2)][int(rs.randint(4))]
    P = int(rs.choice([1, 1, 2, 4])) if n - 2 >= B else 1
    _cfg(monkeypatch, B, logR, int(rs.choice([0, 1, 2])))
    omsc, arrs = _orc_msc(H)
    x = _rand(1 << n, seed)
    ref = orc.matvec_general(omsc, osub, osub, x)
    tol = 64 * len(arrs[2]) * EPS * max(1.0, np.abs(arrs[3]).max()) * np.abs(x).max()
    nloc = (1 << n) // P
    seen = {"tabs": 0, "groups": 0}
    for on in ("1", "0"):
        monkeypatch.setenv("DNM_TAB_RECORDS", on)
        monkeypatch.setenv("DNM_DIAG_GROUPS", on)
        y = np.zeros(1 << n, dtype=complex)
        for r in range(P):
            hm = HostMat(*arrs, sub._c(), sub._c(), rank=r, nranks=P)
            assert hm.tiled == 1, hm.describe()
            yl = np.zeros(nloc, dtype=complex)
            for p in hm.local:
                run_pass(hm, p, x[r * nloc:(r + 1) * nloc], yl)
            for i, (partner, off, cnt) in enumerate(hm.recvs):
                run_remote(hm, i, x[partner * nloc + off:partner * nloc + off + cnt], yl)
            y[r * nloc:(r + 1) * nloc] = yl
            for desc, quads in hm.local + hm.remote:
                if on == "1":
                    seen["tabs"] += desc.tab_loop[2]
                    seen["groups"] += desc.gbucket[_lib.MAXR] - desc.gbucket[0]
                else:
                    assert desc.tab_loop[2] == 0 and desc.gbucket[_lib.MAXR] == desc.gbucket[0]
        assert np.max(np.abs(y - ref)) <= tol, (on, hm.describe())
    assert seen["tabs"] > 0, "no mask of this operator took the table form"


@pytest.mark.parametrize("B,logR,mode,P", [(8, 2, 2, 1), (10, 3, 2, 1), (8, 2, 0, 1), (8, 2, 2, 4)])
def test_grouped_diagonal_terms(monkeypatch, B, logR, mode, P):
    """Diagonal terms that see the tile and bits outside it (an all-to-all ZZ coupling: benchmark.py's long_range), grouped by
    their sign mask inside the tile (DevPass::gbucket): a group's sum over the outside bits once per workgroup, one term per
    group for the threads.  The emulation of the exported records against the oracle, whole and on four ranks (rank bits
    among the outside bits); DNM_DIAG_GROUPS=0 lists every term and gives the same product."""
    L = 14
    _cfg(monkeypatch, B, logR, mode)
    H = models.long_range(L)
    omsc, arrs = _orc_msc(H)
    sub = Full(L=L)
    x = _rand(1 << L, 9)
    ref = orc.matvec_general(omsc, orc.full(L), orc.full(L), x)
    tol = 64 * len(arrs[0]) * EPS * np.abs(arrs[3]).max() * np.abs(x).max() * L
    nloc = (1 << L) // P
    for on in (True, False):
        monkeypatch.setenv("DNM_DIAG_GROUPS", "1" if on else "0")
        y = np.zeros(1 << L, dtype=complex)
        ngroups = nlisted = 0
        for r in range(P):
            hm = HostMat(*arrs, sub._c(), sub._c(), rank=r, nranks=P)
            yl = np.zeros(nloc, dtype=complex)
            for p in hm.local:
                run_pass(hm, p, x[r * nloc:(r + 1) * nloc], yl)
            for i, (partner, off, cnt) in enumerate(hm.recvs):
                run_remote(hm, i, x[partner * nloc + off:partner * nloc + off + cnt], yl)
            y[r * nloc:(r + 1) * nloc] = yl
            for desc, quads in hm.local:
                if desc.has_diag:
                    ngroups += desc.gbucket[_lib.MAXR] - desc.gbucket[0]
                    nlisted += desc.dbucket[_lib.MAXR] - desc.dbucket[0]
                else:
                    assert desc.gbucket[_lib.MAXR] == desc.gbucket[0]
        assert np.max(np.abs(y - ref)) <= tol
        if on:
            # every spin of the tile couples to every spin outside it: one group per tile spin (and rank)
            assert ngroups == P * B, (ngroups, hm.describe())
            listed_on = nlisted
        else:
            assert ngroups == 0 and nlisted > listed_on


@pytest.mark.parametrize("spaces", [(0, 0), (1, 1), (0, 1), (1, 0)])
@pytest.mark.parametrize("name", ["long_range", "ising", "mbl", "syk"])
def test_tiled_plan_parity(monkeypatch, name, spaces):
    L = 12 if name != "syk" else 10         # (syk: table records with the dropped spin folded into the sign masks)
    if name == "syk" and spaces[0] != spaces[1]:
        pytest.skip("SYK conserves the parity: nothing maps one sector onto the other")
    _cfg(monkeypatch, 8, 2)
    H = models.BY_NAME[name](L)
    omsc, arrs = _orc_msc(H)
    left, right = Parity(spaces[0], L=L), Parity(spaces[1], L=L)
    hm = HostMat(*arrs, left._c(), right._c())
    assert hm.tiled == 1
    x = _rand(1 << (L - 1), 1)
    y = multiply(hm, x)
    ref = orc.matvec_general(omsc, orc.parity(L, spaces[0]), orc.parity(L, spaces[1]), x)
    tol = 64 * len(arrs[0]) * EPS * np.abs(arrs[3]).max() * np.abs(x).max()
    assert np.max(np.abs(y - ref)) <= tol


def test_plan_shape_chain_L30(monkeypatch):
    """The headline operator: passes cover every mask exactly once."""
    for B, want in ((12, None), (13, None)):
        _cfg(monkeypatch, B, 4)
        H = models.mbl(30)
        _, arrs = _orc_msc(H)
        sub = Full(L=30)
        hm = HostMat(*arrs, sub._c(), sub._c())
        total = sum(p[0].loop[_lib.LP_COUNT] - p[0].loop[0] for p in hm.local)
        assert total == len(arrs[0]) - 1          # one record per off-diagonal mask
        assert sum(p[0].has_diag for p in hm.local) == 1
        assert hm.local[0][0].accumulate == 0 and all(p[0].accumulate for p in hm.local[1:])
        seen = set()
        for desc, quads in hm.local:
            assert desc.loop[5] == desc.loop[_lib.LP_COUNT]       # no gathers
            for M in quads[desc.loop[0]:desc.loop[5]]:
                # recover the global mask from its tile coordinates
                g = 0
                for j in range(desc.nseg):
                    seg = (M.mask_tile >> desc.seg_off[j]) & ((1 << desc.seg_len[j]) - 1)
                    g |= seg << desc.seg_pos[j]
                seen.add(g)
        assert seen == set(int(m) for m in arrs[0][1:])


@pytest.mark.parametrize("model,P", [("ising", 2), ("ising", 4), ("long_range", 4), ("xsum", 8)])
def test_partitioned_plan_other_models(monkeypatch, model, P):
    """Operators whose rank-bit masks never vanish (single-spin flips, long-range XX)
    take both halves of every partner block."""
    L = 13
    _cfg(monkeypatch, 8, 2)
    H = models.BY_NAME[model](L)
    omsc, arrs = _orc_msc(H)
    sub = Full(L=L)
    x = _rand(1 << L, 5)
    nloc = (1 << L) // P
    y = np.zeros(1 << L, dtype=complex)
    for r in range(P):
        hm = HostMat(*arrs, sub._c(), sub._c(), rank=r, nranks=P)
        yl = np.zeros(nloc, dtype=complex)
        for p in hm.local:
            run_pass(hm, p, x[r * nloc:(r + 1) * nloc], yl)
        for i, (partner, off, cnt) in enumerate(hm.recvs):
            run_remote(hm, i, x[partner * nloc + off:partner * nloc + off + cnt], yl)
        y[r * nloc:(r + 1) * nloc] = yl
    ref = orc.matvec(omsc, orc.full(L), orc.full(L), x)
    assert np.max(np.abs(y - ref)) <= 64 * len(arrs[0]) * EPS * np.abs(x).max() * max(1.0, np.abs(arrs[3]).max())


@pytest.mark.parametrize("S", [5, 7, 9])
@pytest.mark.parametrize("P", [1, 2, 4])
@pytest.mark.parametrize("model", ["mbl", "ising"])
def test_swizzled_layout_plan(monkeypatch, model, P, S):
    """Vectors in the XOR-swizzled layout (dnm_subspace.vec_swizzle = S): the pass tables carry the shift and, for
    partner passes, the constants of the sub-block offsets; local blocks are device images, the exchange ships
    slices of them as they lie.  Result, un-swizzled, must equal the oracle's."""
    from dynamite_amd.config import config
    from plan_emulator import vec_pos
    monkeypatch.setattr(config, "vec_swizzle", S)
    L = 14
    _cfg(monkeypatch, 8, 2, 2)
    monkeypatch.setenv("DNM_GBITS", "3")
    H = models.BY_NAME[model](L)
    omsc, arrs = _orc_msc(H)
    sub = Full(L=L)
    assert sub._c().vec_swizzle == S
    x = _rand(1 << L, 9)
    nloc = (1 << L) // P
    pos = vec_pos(np.arange(nloc), S)
    assert not np.array_equal(pos, np.arange(nloc))
    dev = []                                     # every rank's block as it lies in device memory
    for r in range(P):
        img = np.empty(nloc, dtype=complex)
        img[pos] = x[r * nloc:(r + 1) * nloc]
        dev.append(img)
    y = np.zeros(1 << L, dtype=complex)
    for r in range(P):
        hm = HostMat(*arrs, sub._c(), sub._c(), rank=r, nranks=P)
        assert all(p[0].swz_shift == S for p in hm.local + hm.remote)
        yl = np.full(nloc, np.nan + 0j)
        for p in hm.local:
            run_pass(hm, p, dev[r], yl)
        for i, (partner, off, cnt) in enumerate(hm.recvs):
            run_remote(hm, i, dev[partner][off:off + cnt], yl)
        y[r * nloc:(r + 1) * nloc] = yl[pos]
    ref = orc.matvec(omsc, orc.full(L), orc.full(L), x)
    assert np.max(np.abs(y - ref)) <= 64 * len(arrs[0]) * EPS * np.abs(x).max() * max(1.0, np.abs(arrs[3]).max())


@pytest.mark.parametrize("P", [2, 4, 8])
def test_partitioned_plan_matches_oracle_ranks(monkeypatch, P):
    """Rank-local + partner passes reproduce the reference's multi-rank Fast
    path semantics (emulated ranks, exchange = numpy slicing)."""
    L = 15
    _cfg(monkeypatch, 8, 2)
    H = models.mbl(L)
    omsc, arrs = _orc_msc(H)
    sub = Full(L=L)
    x = _rand(1 << L, 2)
    nloc = (1 << L) // P
    y = np.zeros(1 << L, dtype=complex)
    partners_seen = set()
    plans = {}
    for r in range(P):
        hm = HostMat(*arrs, sub._c(), sub._c(), rank=r, nranks=P)
        yl = np.zeros(nloc, dtype=complex)
        xl = x[r * nloc:(r + 1) * nloc]
        for p in hm.local:
            run_pass(hm, p, xl, yl)
        assert len(hm.remote) == len(hm.recvs)
        for i, (partner, off, cnt) in enumerate(hm.recvs):
            assert partner != r and 0 <= partner < P
            partners_seen.add((r, partner))
            run_remote(hm, i, x[partner * nloc + off:partner * nloc + off + cnt], yl)
        plans[r] = (hm.sends, hm.recvs)
        y[r * nloc:(r + 1) * nloc] = yl
    osub = orc.full(L)
    ref = orc.matvec_fast_ranks(omsc, osub, x, P)
    tol = 64 * len(arrs[0]) * EPS * np.abs(x).max()
    assert np.max(np.abs(y - ref)) <= tol
    # chain: partners are r^1 (P>=2), r^3 (P>=4), r^6 (P=8) -- XOR images of masks 3<<i
    want = {1} | ({3, 2} if P >= 4 else set()) | ({6, 4} if P >= 8 else set())
    hs = {a ^ b for a, b in partners_seen}
    assert hs <= {1, 2, 3, 4, 6} and 1 in hs
    # the schedules agree pairwise: r's sends to q, in order, are q's receives from r
    for r in range(P):
        for q in range(P):
            assert [(o, c) for p_, o, c in plans[r][0] if p_ == q] == [(o, c) for p_, o, c in plans[q][1] if p_ == r]
    # flip-flop bonds: the boundary bond moves half a block, a bond inside the rank bits a whole
    # block but only between ranks whose two bits differ -- never more than the old whole-block exchange
    total = sum(c for r in range(P) for _, _, c in plans[r][1])
    full = sum(len({p_ for p_, _, _ in plans[r][1]}) for r in range(P)) * nloc
    assert total < full or P == 1
    if P == 2:
        assert total == nloc


# ------------------------------------------------------------------ XParity host side

def test_xparity_reduce_msc_golden(golden_xp):
    """dynamite_amd.subspaces.XParity.reduce_msc against the reference's
    (subspaces.py:632-674) on every fixture, plus its validation rules (:566-617)."""
    from dynamite_amd.subspaces import XParity, Explicit
    for name in golden_xp.names():
        g = golden_xp[name]
        L, sector = int(g["L"]), int(g["sector"])
        msc = np.zeros(g["in_masks"].size, dtype=msc_tools.msc_dtype)
        msc["masks"], msc["signs"], msc["coeffs"] = g["in_masks"], g["in_signs"], g["in_coeffs"]
        if "_sc" in name:
            parent = SpinConserve(L, L // 2)
        elif "parity" in name:
            parent = Parity("even" if "even" in name else "odd", L=L)
        else:
            parent = Full(L=L)
        sub = XParity(parent, sector=sector)
        red, conserved = sub.reduce_msc(msc, check_conserves=True)
        assert conserved == bool(g["conserved"]), name
        masks, offs = msc_tools.get_mask_offsets(red)
        assert np.array_equal(masks, g["masks"]) and np.array_equal(offs, g["mask_offsets"]), name
        assert np.array_equal(red["signs"], g["signs"]) and np.array_equal(red["coeffs"], g["coeffs"]), name
        assert sub.get_dimension() == g["x"].size
        assert not sub.product_state_basis and sub.L == L
        assert sub.idx_to_state(3) == parent.idx_to_state(3)
        with pytest.raises(ValueError):
            sub.state_to_idx(1 << (L - 1))
    with pytest.raises(ValueError):
        XParity(Parity("even", L=7))
    with pytest.raises(ValueError):
        XParity(SpinConserve(8, 3))
    with pytest.raises(ValueError):
        XParity(Full(L=6), sector=2)
    with pytest.raises(ValueError):
        XParity(XParity(Full(L=6)))
    XParity(Explicit([0, 1, 6, 7], L=3))                     # closed under the flip, first half has spin 2 up
    with pytest.raises(ValueError):
        XParity(Explicit([0, 1, 2, 7], L=3))                 # complement of 1 (=6) missing
    with pytest.raises(ValueError):
        XParity(Explicit([0, 7, 1, 6], L=3))                 # a first-half state has spin 2 down
    assert XParity(Full(L=6), "+") == XParity(Full(L=6), +1)
    assert XParity(Full(L=6), "+") != XParity(Full(L=6), "-")
    assert hash(XParity(Full(L=6), "-")) == hash(XParity(Full(L=6), -1))


# ------------------------------------------------------------------ entropies (host numpy, tests/unit/test_entropies.py)

def test_entropy_functions(known):
    from conftest import cmatrix
    from dynamite_amd.computations import dm_entanglement_entropy, dm_renyi_entropy
    e = known["entropies"]
    for c in e["von_neumann"]:
        assert abs(dm_entanglement_entropy(cmatrix(c["dm"])) - c["value"]) < 1e-5     # 6-digit tables
    for c in e["cases"]:
        dm = cmatrix(c["dm"])
        for alpha, val in c["renyi"]:
            assert abs(dm_renyi_entropy(dm, alpha, 'eigsolve') - val) < 1e-14
            if alpha == 'inf' or int(alpha) == alpha:
                assert abs(dm_renyi_entropy(dm, alpha, 'matrix_power') - val) < 1e-14
            else:
                with pytest.raises(TypeError):
                    dm_renyi_entropy(dm, alpha, 'matrix_power')
        assert abs(dm_entanglement_entropy(dm) - dict((str(a), v) for a, v in c["renyi"])["1"]) < 1e-14
    with pytest.raises(ValueError):
        dm_renyi_entropy(np.eye(2) / 2, 2, 'bogus')


def test_auto_subspace():
    """Auto (subspaces.py:465-530): the component of a state under H.  XX+YY from a half-filled
    state gives the SpinConserve sector (tests/unit/test_subspaces.py Auto cases in spirit);
    sort=False is the reversed serial breadth-first order of bsubspace.pyx:212-261."""
    from dynamite_amd.subspaces import Auto
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, index_sum
    L = 8
    H = index_sum(sigmax(0) * sigmax(1) + sigmay(0) * sigmay(1), size=L)
    a = Auto(H, 'U' * 4 + 'D' * 4)
    sc = SpinConserve(L, 4)
    assert np.array_equal(a.state_map, sc.idx_to_state(np.arange(sc.get_dimension())))
    assert a == sc and a.get_dimension() == 70
    b = Auto(H, 'U' * 4 + 'D' * 4, sort=False)
    # serial reference search
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    order, seen, i = [0b11110000], {0b11110000}, 0
    while i < len(order):
        st = order[i]
        for mi in range(masks.size):
            tot = sum((1 - 2 * (bin(st & int(H.msc['signs'][t])).count('1') & 1)) * H.msc['coeffs'][t]
                      for t in range(offs[mi], offs[mi + 1]))
            e = st ^ int(masks[mi])
            if tot != 0 and e not in seen:
                seen.add(e); order.append(e)
        i += 1
    assert b.state_map.tolist() == order[::-1]
    assert sorted(b.state_map.tolist()) == a.state_map.tolist()
    # an operator that connects everything: the full space; too small a guess is refused
    X = index_sum(sigmax(), size=5)
    assert Auto(X, 0).get_dimension() == 32
    with pytest.raises(RuntimeError):
        Auto(X, 0, size_guess=10)
    Z = index_sum(sigmaz(), size=5)
    assert Auto(Z, 'DUDUU').state_map.tolist() == [0b00101]


def test_tools_host(capsys):
    """dynamite_amd.tools (reference tools.py): rank-aware print and the fixed build facts."""
    from dynamite_amd import tools
    tools.mpi_print("hello", rank=0)
    tools.mpi_print("silent", rank=1)
    assert capsys.readouterr().out == "hello\n"
    assert tools.complex_enabled()
    with pytest.raises(ValueError):
        tools.get_max_memory_usage(which='rank')


def test_global_config_L():
    """config.L set globally (how the reference's scripts and benchmark.py work): operator algebra,
    copies and default subspaces pick it up."""
    from dynamite_amd import config
    from dynamite_amd.operators import sigmax, sigmaz, index_sum, Operator
    old = config.L
    try:
        config.L = 9
        H = index_sum(0.25 * sigmaz(0) * sigmaz(1)) + 0.1 * index_sum(sigmax())
        assert H.L == 9 and H.copy().L == 9 and Operator().L == 9
        assert H.max_spin_idx == 8 and H.nterms == 8 + 9
        assert (2 * H).L == 9 and H.left_subspace.L == 9
        with pytest.raises(ValueError):
            sigmax(9).L = 9
    finally:
        config.L = old


def test_window_needed_ranges_and_schedule():
    """backend.needed_ranges turns the device's chunk map into column ranges; window_exchange_ops schedules only
    those, consistently on every rank (what q sends to r is what r expects from q, in the same order)."""
    from dynamite_amd.backend import needed_ranges, window_exchange_ops, split_ownership
    cmap = np.zeros(10, dtype=np.uint8)
    cmap[[0, 1, 2, 5, 9]] = 1
    assert needed_ranges(cmap, 4, (37, 187)) == [(37, 80), (112, 128), (176, 188)]
    assert needed_ranges(np.zeros(4, dtype=np.uint8), 3, (8, 30)) == []
    assert needed_ranges(np.ones(3, dtype=np.uint8), 0, (5, 7)) == [(5, 8)]
    rs = np.random.RandomState(3)
    N, P = 5000, 5
    owned = [split_ownership(N, P, q) for q in range(P)]
    windows, needs = [], []
    for q, (s0, n) in enumerate(owned):
        lo, hi = max(0, s0 - rs.randint(0, 1500)), min(N - 1, s0 + n - 1 + rs.randint(0, 1500))
        windows.append((lo, hi))
        cuts = np.sort(rs.choice(np.arange(lo, hi + 2), size=6, replace=False))
        needs.append([(int(cuts[i]), int(cuts[i + 1])) for i in (0, 2, 4)])
    ops = [window_exchange_ops(owned, windows, q, needs) for q in range(P)]
    for q in range(P):
        recvs, sends = ops[q]
        for r in range(P):
            if r == q:
                continue
            assert [(lo, hi) for (src, lo, hi) in recvs if src == r] == \
                   [(lo, hi) for (dst, lo, hi) in ops[r][1] if dst == q]
        got = sorted((lo, hi) for _, lo, hi in recvs)
        # everything needed that others own arrives, nothing else
        want = []
        for a, b in needs[q]:
            for r, (r0, rn) in enumerate(owned):
                if r != q and max(a, r0) < min(b, r0 + rn):
                    want.append((max(a, r0), min(b, r0 + rn)))
        assert got == sorted(want)
    # without the ranges the whole window travels
    full = window_exchange_ops(owned, windows, 2)
    assert sum(hi - lo for _, lo, hi in full[0]) == windows[2][1] + 1 - windows[2][0] - \
        (min(windows[2][1] + 1, owned[2][0] + owned[2][1]) - max(windows[2][0], owned[2][0]))


@pytest.mark.parametrize("L,P,S,flags_amin", [(30, 1, 16, 0), (31, 1, 16, 0), (32, 1, 16, 0), (34, 8, 14, 0), (34, 8, 14, 8),
                                              (33, 4, 14, 7), (26, 1, 16, 0), (20, 1, 0, 0), (24, 3, 0, 0)])
def test_pass_addressing_fits_32_bit_offsets(L, P, S, flags_amin):
    """tile_pass_kernel addresses a row as (scalar base) + (32-bit byte offset of the thread): every position bit the
    thread's part of the tile coordinate can reach -- the low B - LOGR tile bits and what the layout folds them onto --
    lies in DevPass::pos_tmask, below bit 28, for every pass of the plans the benchmark configurations take (a tile
    that reaches higher, like the layout-B operator of the transposed exchange at 2^31 amplitudes per rank, gets
    more rows per thread: csrc/mat.cpp build_pass)."""
    from dynamite_amd import models, msc_tools, _lib
    from dynamite_amd.subspaces import Full
    from plan_emulator import HostMat, vec_pos
    H = models.heisenberg(L)
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    sub = Full(L=L)
    c = sub._c()
    pow2 = P & (P - 1) == 0
    c.vec_swizzle = S if pow2 else 0
    for rank in sorted({0, P - 1, P // 2}):
        flags = _lib.MAT_HOST_ONLY | (flags_amin << _lib.MAT_AMIN_SHIFT)
        hm = HostMat(masks, offs, H.msc['signs'], H.msc['coeffs'], c, c, rank=rank, nranks=P, flags=flags)
        if not hm.tiled:
            continue
        for desc, _ in hm.local + hm.remote:
            lognt = desc.tile_bits - desc.log_rows
            want, cnt = 0, 0
            for j in range(desc.nseg):
                for i in range(desc.seg_len[j]):
                    if cnt < lognt:
                        bit = 1 << (desc.seg_pos[j] + i)
                        want |= int(vec_pos(bit, desc.swz_shift)) | bit
                    cnt += 1
            assert desc.pos_tmask == want, (L, P, rank, hm.describe())
            assert desc.pos_tmask >> 28 == 0


def test_complement_ranges():
    """The rows a window partition multiplies after its exchange are what the listed local ranges leave out."""
    from dynamite_amd.backend import complement_ranges
    assert complement_ranges([], 10) == [(0, 10)]
    assert complement_ranges([(0, 10)], 10) == []
    assert complement_ranges([(2, 4), (4, 6), (8, 9)], 10) == [(0, 2), (6, 8), (9, 10)]
    rs = np.random.RandomState(3)
    for _ in range(50):
        n = int(rs.randint(1, 1000))
        cuts = np.unique(rs.randint(0, n + 1, size=int(rs.randint(0, 12))))
        ranges = [(int(a), int(b)) for a, b in zip(cuts[0::2], cuts[1::2])]
        rest = complement_ranges(ranges, n)
        cover = np.zeros(n, dtype=int)
        for a, b in ranges + rest:
            cover[a:b] += 1
        assert np.all(cover == 1)
    with pytest.raises(ValueError):
        complement_ranges([(4, 2)], 10)
    with pytest.raises(ValueError):
        complement_ranges([(0, 5), (3, 8)], 10)


@pytest.mark.parametrize("name,L,sub", [("mbl", 14, "full"), ("xxz", 13, "full"), ("heisenberg", 14, "full"),
                                        ("localized", 13, "full"), ("ising", 14, "full"), ("ising", 15, "parity0"),
                                        ("ising", 15, "parity1"), ("xsum", 13, "full")])
def test_real_packed_operator_form(monkeypatch, name, L, sub):
    """DNM_MAT_REAL_PACKED (csrc/mat.cpp pack_opform): the records of a real-symmetric operator in real arithmetic --
    vectors of dim / 2 elements holding two real amplitudes each -- run through the kernel emulation reproduce the
    oracle's y = H x for a real x; an operator with an imaginary matrix element has no such form."""
    from dynamite_amd import models, msc_tools, _lib
    from dynamite_amd.subspaces import Full, Parity
    from oracle import oracle as orc
    from plan_emulator import HostMat, multiply, vec_pos
    for k, v in (("DNM_TILE_BITS", "8"), ("DNM_LOG_ROWS", "2"), ("DNM_PLAN_MODE", "2"), ("DNM_GBITS", "3"), ("DNM_AMIN", "3")):
        monkeypatch.setenv(k, v)
    H = models.BY_NAME[name](L)
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    s = Full(L=L) if sub == "full" else Parity(int(sub[-1]), L=L)
    c = s._c()
    c.vec_swizzle = 6
    hm = HostMat(masks, offs, H.msc['signs'], H.msc['coeffs'], c, c, flags=_lib.MAT_REAL_PACKED)
    dim = s.get_dimension()
    assert hm.tiled and (1 << hm.n_loc) == dim // 2
    assert all(d.cache_policy & 256 for d, _ in hm.local)
    rs = np.random.RandomState(L)
    xr = rs.standard_normal(dim)
    pos = vec_pos(np.arange(dim // 2), 6)
    xp = np.empty(dim // 2, dtype=np.complex128)
    xp[pos] = xr[0::2] + 1j * xr[1::2]                # element j = amplitudes 2j and 2j + 1, in the vector layout
    yp = multiply(hm, xp)[pos]
    y = np.empty(dim)
    y[0::2], y[1::2] = yp.real, yp.imag
    osub = orc.full(L) if sub == "full" else orc.parity(L, int(sub[-1]))
    ref = orc.matvec(orc.Msc(masks, offs, H.msc['signs'], H.msc['coeffs']), osub, osub, xr.astype(np.complex128))
    assert np.abs(ref.imag).max() == 0.0
    assert np.abs(y - ref.real).max() <= 1e-13 * max(1.0, np.abs(ref).max()), hm.describe()


def test_real_packed_refuses_imaginary_elements():
    from dynamite_amd import msc_tools, _lib, backend
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, index_sum
    from dynamite_amd.subspaces import Full
    L = 13
    H = index_sum(sigmax(0) * sigmay(1) - sigmay(0) * sigmax(1), size=L) + index_sum(sigmaz(0), size=L)
    H.L = L
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    c = Full(L=L)._c()
    with pytest.raises(_lib.BackendError, match="imaginary"):
        backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], c, c, False,
                           _lib.MAT_HOST_ONLY | _lib.MAT_REAL_PACKED, 0, 1)


@pytest.mark.parametrize("P", [2, 4, 8])
def test_real_packed_partitioned_plan(monkeypatch, P):
    """A real-packed operator on 2^p ranks: the rank bits are the top index bits and the packed bit is bit 0, so the
    partner exchange is that of an operator on one bit less -- rank-local and partner passes (emulated ranks, exchange
    = numpy slicing of the packed vector) reproduce the oracle for a real x."""
    from dynamite_amd import _lib
    L = 15
    _cfg(monkeypatch, 8, 2)
    H = models.mbl(L)
    omsc, arrs = _orc_msc(H)
    sub = Full(L=L)
    xr = np.random.RandomState(9).standard_normal(1 << L)
    xp = xr[0::2] + 1j * xr[1::2]
    nloc = (1 << (L - 1)) // P
    yp = np.zeros(1 << (L - 1), dtype=complex)
    for r in range(P):
        hm = HostMat(*arrs, sub._c(), sub._c(), rank=r, nranks=P, flags=_lib.MAT_REAL_PACKED)
        assert hm.tiled and (1 << hm.n_loc) == nloc and hm.recvs
        yl = np.zeros(nloc, dtype=complex)
        for p in hm.local:
            run_pass(hm, p, xp[r * nloc:(r + 1) * nloc], yl)
        for i, (partner, off, cnt) in enumerate(hm.recvs):
            run_remote(hm, i, xp[partner * nloc + off:partner * nloc + off + cnt], yl)
        yp[r * nloc:(r + 1) * nloc] = yl
    y = np.empty(1 << L)
    y[0::2], y[1::2] = yp.real, yp.imag
    ref = orc.matvec(omsc, orc.full(L), orc.full(L), xr.astype(complex)).real
    assert np.max(np.abs(y - ref)) <= 64 * len(arrs[0]) * EPS * np.abs(xr).max()
