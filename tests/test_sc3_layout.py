"""
Host logic of the SpinConserve internal vector layout (dynamite_amd/csrc/sc3.h) -- no GPU needed: the position map
is a bijection onto the non-padding slots that keeps the reference order (bsubspace_impl.h:187-245) at the level of
the top-bit blocks and inside every row, partitions hand whole blocks to the ranks, and a host-only handle of
BASELINE config 5 (SpinConserve(36, 18) on 8 ranks) plans the two tiled passes and its column windows.
"""
import ctypes as C
import math

import numpy as np
import pytest

from dynamite_amd import _lib, backend, models, msc_tools
from dynamite_amd.subspaces import SpinConserve


def _desc(L, k, a, w):
    sub = SpinConserve(L, k)
    d = _lib.Subspace.from_buffer_copy(sub._c())
    d.vec_swizzle = a | (w << 8)
    return sub, d


def _positions(d, n, part=None):
    idx = np.arange(n, dtype=np.int64)
    pos = np.empty_like(idx)
    _lib.check(_lib.lib().dnm_vec_layout_positions_host(C.byref(d), C.byref(part) if part is not None else None, n,
                                                        _lib.p64(idx), _lib.p64(pos)))
    return pos


@pytest.mark.parametrize("L,k,a,w", [(11, 5, 6, 4), (12, 6, 6, 4), (13, 3, 6, 4), (14, 7, 5, 4), (12, 0, 6, 4), (12, 12, 6, 4)])
def test_positions_are_a_bijection_that_keeps_rows_together(L, k, a, w):
    sub, d = _desc(L, k, a, w)
    n = sub.get_dimension()
    nint = C.c_int64()
    _lib.check(_lib.lib().dnm_vec_layout_size(C.byref(d), C.byref(nint)))
    pos = _positions(d, n)
    assert len(np.unique(pos)) == n and pos.min() >= 0 and pos.max() < nint.value
    assert nint.value >= n and nint.value % 8 == 0
    states = sub.idx_to_state(np.arange(n))
    # inside a row (equal bits above Lo) consecutive states sit at consecutive positions, rows start on 128-byte lines
    hi = states >> a
    same_row = hi[1:] == hi[:-1]
    assert np.all(pos[1:][same_row] == pos[:-1][same_row] + 1)
    first = np.concatenate(([True], ~same_row))
    assert np.all(pos[first] % 8 == 0)
    # blocks of equal top bits stay in the reference's order and are contiguous
    T = states >> (a + w)
    for t in np.unique(T):
        p = pos[T == t]
        later = pos[T > t]
        assert later.size == 0 or p.max() < later.min()
    # inside a block the rows are grouped by popcount(W), ascending
    W = (states >> a) & ((1 << w) - 1)
    cw = np.array([bin(int(v)).count('1') for v in W])
    for t in np.unique(T):
        sel = T == t
        order = np.argsort(pos[sel], kind='stable')
        assert np.all(np.diff(cw[sel][order]) >= 0)


@pytest.mark.parametrize("P", [2, 3, 8])
def test_partition_gives_whole_blocks(P):
    L, k, a, w = 14, 7, 6, 4
    sub, d = _desc(L, k, a, w)
    n = sub.get_dimension()
    nint = C.c_int64()
    _lib.check(_lib.lib().dnm_vec_layout_size(C.byref(d), C.byref(nint)))
    states = sub.idx_to_state(np.arange(n))
    gpos = _positions(d, n)
    i_end = n_end = 0
    for r in range(P):
        istart, ilen, nstart, nlen = backend.layout_partition(d, P, r)
        assert istart == i_end and nstart == n_end          # contiguous in both index spaces
        i_end, n_end = istart + ilen, nstart + nlen
        if nlen:
            T = states[nstart:nstart + nlen] >> (a + w)
            others = np.concatenate((states[:nstart], states[nstart + nlen:])) >> (a + w)
            assert not np.intersect1d(T, others).size       # whole blocks of equal top bits
            part = _lib.Partition(r, P)
            lp = _positions(d, nlen, part)
            assert np.array_equal(lp + istart, gpos[nstart:nstart + nlen])
    assert i_end == nint.value and n_end == n


def test_config5_plan_on_the_host():
    """SpinConserve(36, 18) on 8 ranks (BASELINE configs[4]) in the (14, 10) layout: 0.2 % padding, ranks balanced to a
    block, the two tiled passes planned, and what a rank reads beyond its own part."""
    L, k, a, w, P = 36, 18, 14, 10, 8
    sub, d = _desc(L, k, a, w)
    dim = math.comb(L, k)
    nint = C.c_int64()
    _lib.check(_lib.lib().dnm_vec_layout_size(C.byref(d), C.byref(nint)))
    assert dim < nint.value < dim * 1.004
    H = models.heisenberg(L)
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    parts = [backend.layout_partition(d, P, r) for r in range(P)]
    lens = [p[1] for p in parts]
    assert max(lens) - min(lens) < 0.01 * nint.value / P
    assert sum(p[3] for p in parts) == dim
    for r in (0, 3, 7):
        h = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], d, d, False, _lib.MAT_HOST_ONLY, r, P)
        buf = C.create_string_buffer(4096)
        _lib.check(_lib.lib().dnm_mat_plan_describe(h, buf, len(buf)))
        s = buf.value.decode()
        assert "two-pass" in s and "[T 12 | W 10 | Lo 14]" in s and "13 gathered" not in s
        row0, ml = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().dnm_mat_ownership(h, C.byref(row0), C.byref(ml)))
        assert (row0.value, ml.value) == parts[r][:2]
        lo, hi = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().dnm_mat_column_window(h, C.byref(lo), C.byref(hi), None))
        assert lo.value <= row0.value and hi.value >= row0.value + ml.value - 1
        assert hi.value - lo.value + 1 <= nint.value
        _lib.check(_lib.lib().dnm_mat_destroy(h))


def test_layout_choice_depends_on_rank_count():
    """The internal layout hands whole top-bit blocks to a rank, 2^(L - 24) of them under (14, 10): it is taken on
    several ranks only where no rank stays empty and the largest share is within 10 % of the mean
    (SpinConserve.layout_usable); elsewhere vectors stay in reference order, split as PETSc splits them."""
    from dynamite_amd.subspaces import SpinConserve
    from dynamite_amd import backend, _lib
    import dynamite_amd.subspaces as S
    usable = {(26, 13, 4): False, (26, 13, 8): False, (27, 13, 8): False, (28, 14, 8): False, (28, 14, 2): True,
              (32, 16, 8): True, (34, 17, 8): True, (36, 18, 8): True, (36, 18, 4): True, (30, 15, 4): True,
              (26, 13, 1): True}
    for (L, k, P), want in usable.items():
        sub = SpinConserve(L, k)
        assert sub.layout_usable(14, 10, P) == want, (L, k, P)
        if P > 1:
            d = _lib.Subspace()
            d.type, d.L, d.k, d.ld_nchoosek = S.SPIN_CONSERVE, L, k, L + 1
            d.nchoosek = _lib.p64(sub._nchoosek)
            d.vec_swizzle = 14 | (10 << 8)
            rows = [backend.layout_partition(d, P, q)[3] for q in range(P)]
            assert sum(rows) == sub.get_dimension()
            assert want == (min(rows) > 0 and max(rows) <= 1.10 * sum(rows) / P)
    # one rank: the production choice is the internal layout from 2^22 states on
    assert SpinConserve(26, 13).vec_swizzle == (14 | (10 << 8)) and SpinConserve(24, 12).vec_swizzle == 0


# ---- site relabelling (dnm_subspace.site_perm, csrc/sc3_perm.cpp) -- host logic ---------------------------------

def _kagome_masks(name):
    from dynamite_amd import lattices
    n, edges = lattices.kagome(name)
    return n, np.array(sorted((1 << i) | (1 << j) for i, j in edges), dtype=np.int64)


def test_lattice_generator_matches_the_reference_edge_lists():
    """dynamite_amd/lattices.py against the edge lists the reference's lattice_library.basis_to_graph produces for every
    cluster of its library (tests/golden/kagome_edges.json, generated by make_golden.py): same vertex numbering."""
    import json
    import os
    from dynamite_amd import lattices
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kagome_edges.json")))
    assert set(ref) == set(lattices.KAGOME_CLUSTERS)
    for name, d in ref.items():
        n, edges = lattices.kagome(name)
        assert n == d["n"] and [list(e) for e in edges] == d["edges"], name
        deg = np.bincount(np.array(edges).ravel(), minlength=n)
        assert np.all(deg == 4)


def test_positions_under_a_site_relabelling():
    """With site_perm the position of reference index i is the plain layout's position of the state with its bits moved
    (spin s -> bit perm[s]); the map stays a bijection onto the non-padding slots."""
    L, k, a, w = 13, 6, 6, 4
    sub, d = _desc(L, k, a, w)
    n = sub.get_dimension()
    plain = _positions(d, n)
    rs = np.random.RandomState(3)
    for _ in range(3):
        perm = rs.permutation(L).astype(np.int8)
        dp = backend.with_site_perm(d, perm)
        assert backend.site_perm_of(dp) == tuple(int(b) for b in perm)
        pos = _positions(dp, n)
        assert len(np.unique(pos)) == n
        states = sub.idx_to_state(np.arange(n))
        moved = np.zeros_like(states)
        for s in range(L):
            moved |= ((states >> s) & 1) << int(perm[s])
        assert np.array_equal(pos, plain[sub.state_to_idx(moved)])
    ident = backend.with_site_perm(d, np.arange(L, dtype=np.int8))
    assert backend.site_perm_of(ident) is None and np.array_equal(_positions(ident, n), plain)
    bad = backend.with_site_perm(d, np.zeros(L, dtype=np.int8))
    with pytest.raises(_lib.BackendError):
        _positions(bad, 4)


def test_site_relabelling_is_chosen_deterministically_and_keeps_chains():
    """dnm_sc_choose_site_perm: a permutation of the spins; the identity for a nearest-neighbour chain (ties keep it,
    so chains keep their kernels); for the kagome tori no more hops between fields than the identity leaves, the same
    answer on every call; under fix_top spin L-1 stays (XParity), also when the masks are the flip-composed ones."""
    chain = np.array([3 << i for i in range(29)], dtype=np.int64)
    perm, counts = backend.choose_site_perm(chain, 30, 14, 10)
    assert np.array_equal(perm, np.arange(30)) and counts == [13, 9, 5, 1, 0, 1]

    def crossing(masks, perm, a=14, w=10):
        f = lambda b: 0 if b < a else (1 if b < a + w else 2)       # noqa: E731
        n = 0
        for m in masks.tolist():
            bits = [b for b in range(64) if (m >> b) & 1]
            n += f(int(perm[bits[0]])) != f(int(perm[bits[1]]))
        return n
    for name in ("27b", "30", "36a"):
        n, masks = _kagome_masks(name)
        perm, counts = backend.choose_site_perm(masks, n, 14, 10)
        assert sorted(perm.tolist()) == list(range(n)) and sum(counts) == len(masks)
        assert counts[3] + counts[4] + counts[5] == crossing(masks, perm) <= crossing(masks, np.arange(n))
        again, _ = backend.choose_site_perm(masks, n, 14, 10)
        assert np.array_equal(perm, again)
    n, masks = _kagome_masks("30")
    perm, counts = backend.choose_site_perm(masks, n, 14, 10, fix_top=True)
    assert perm[n - 1] == n - 1
    # XParity's reduced masks: a hop that touches spin L-1 comes as every spin but the pair (subspaces.py:632-674)
    allm = (1 << n) - 1
    reduced = np.array(sorted(m if not (m >> (n - 1)) & 1 else m ^ allm for m in masks.tolist()), dtype=np.int64)
    perm2, counts2 = backend.choose_site_perm(reduced, n, 14, 10, fix_top=True)
    assert np.array_equal(perm, perm2) and counts == counts2


def test_bond_graph_operator_plans_on_the_host():
    """A host-only handle of the 30-site kagome operator in the relabelled (14, 10) layout: bond-graph passes, the hop
    counts of the chooser; with XParity on top the flip-composed hops are gathered; the long-range model of the
    reference's harness (single-spin fields never act inside the subspace) is a chain with a cached diagonal."""
    H = models.kagome("30")
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    sub, d = _desc(30, 15, 14, 10)
    perm, counts = backend.choose_site_perm(masks, 30, 14, 10)
    dp = backend.with_site_perm(d, perm)
    h = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], dp, dp, False, _lib.MAT_HOST_ONLY, 0, 1)
    buf = C.create_string_buffer(4096)
    _lib.check(_lib.lib().dnm_mat_plan_describe(h, buf, len(buf)))
    desc = buf.value.decode()
    assert "bond graph" in desc and "[T 6 | W 10 | Lo 14]" in desc and "real symmetric" in desc
    assert "%d hops in LDS, %d gathered) then lo pass" % (counts[1], counts[2] + counts[5]) in desc
    assert "%d hops in LDS, %d gathered), diagonal cached" % (counts[0], counts[3] + counts[4]) in desc
    _lib.check(_lib.lib().dnm_mat_destroy(h))
    # XParity: reduce the operator as subspaces.XParity does, spin L-1 fixed
    from dynamite_amd.subspaces import XParity
    xp = XParity(SpinConserve(30, 15), '-')
    red = xp.reduce_msc(H.msc)
    rmasks, roffs = msc_tools.get_mask_offsets(red)
    permx, cx = backend.choose_site_perm(rmasks, 30, 14, 10, fix_top=True)
    dx = backend.with_site_perm(d, permx)
    h = backend.create_mat(rmasks, roffs, red['signs'], red['coeffs'], dx, dx, True, _lib.MAT_HOST_ONLY, 0, 1)
    _lib.check(_lib.lib().dnm_mat_plan_describe(h, buf, len(buf)))
    assert "bond graph" in buf.value.decode()
    M, N, m, n = (C.c_int64() for _ in range(4))
    _lib.check(_lib.lib().dnm_mat_sizes(h, C.byref(M), C.byref(N), C.byref(m), C.byref(n)))
    nint = C.c_int64()
    _lib.check(_lib.lib().dnm_vec_layout_size(C.byref(d), C.byref(nint)))
    assert M.value == math.comb(30, 15) // 2 and m.value == nint.value // 2        # the layout's first half
    _lib.check(_lib.lib().dnm_mat_destroy(h))
    Hl = models.bench_long_range(28)
    Hl.establish_L()
    Hl.reduce_msc()
    lm, lo = msc_tools.get_mask_offsets(Hl.msc)
    _, d28 = _desc(28, 14, 14, 10)
    assert np.array_equal(backend.choose_site_perm(lm, 28, 14, 10)[0], np.arange(28))
    h = backend.create_mat(lm, lo, Hl.msc['signs'], Hl.msc['coeffs'], d28, d28, False, _lib.MAT_HOST_ONLY, 0, 1)
    _lib.check(_lib.lib().dnm_mat_plan_describe(h, buf, len(buf)))
    assert "two-pass kernels, internal layout" in buf.value.decode() and "diagonal cached" in buf.value.decode()
    _lib.check(_lib.lib().dnm_mat_destroy(h))
