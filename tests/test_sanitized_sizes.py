"""
Host logic at the LARGEST sizes the path is specified for, through host-only handles (DNM_MAT_HOST_ONLY: planner,
operator tables, SpinConserve layout tables, exchange plans -- no kernel is launched, no GPU needed): where 32-bit
quantities overflow if they are going to.  Part of the CPU suite; tools/sanitize.sh runs it under AddressSanitizer +
UBSan (profiles/r06_sanitizer.txt).

  * Full space with 2^31 and 2^32 amplitudes per rank (BASELINE configs[3]: L=34 on 8 ranks is n_loc = 31; one GPU
    holds n_loc = 32 in real arithmetic), partner blocks and the transposed exchange split inside the handle;
  * SpinConserve(36, 18) on 8 ranks (BASELINE configs[4]): 9 075 135 300 states, every rank's block of the internal
    three-field layout, its column window and the ranges it reads, positions of sampled indices bit-exact against the
    combinadic rank (bsubspace_impl.h:191-228);
  * the site relabelling of the 36-site kagome torus (run_kagome.py's largest one-GPU cluster).
"""
import ctypes as C
import math

import numpy as np
import pytest

from dynamite_amd import _lib, backend, models, msc_tools
from dynamite_amd.subspaces import Full, Parity, SpinConserve


def _arrays(H):
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    return masks, offs, H.msc['signs'], H.msc['coeffs']


def _describe(h):
    buf = C.create_string_buffer(8192)
    _lib.check(_lib.lib().dnm_mat_plan_describe(h, buf, len(buf)))
    return buf.value.decode()


@pytest.mark.parametrize("L,world,rank,flags", [(34, 8, 5, 0), (35, 8, 7, 0), (35, 16, 9, 0), (32, 1, 0, 0), (31, 1, 0, 0),
                                                (33, 1, 0, _lib.MAT_REAL_PACKED), (36, 8, 3, _lib.MAT_REAL_PACKED)])
def test_full_space_plans_at_31_and_32_local_bits(L, world, rank, flags):
    arrs = _arrays(models.mbl(L) if not flags else models.heisenberg(L))
    sub = Full(L=L)
    lc = _lib.Subspace.from_buffer_copy(sub._c())
    lc.vec_swizzle = 14 if world >= 4 else 16
    h = backend.create_mat(*arrs, lc, lc, False, _lib.MAT_HOST_ONLY | flags, rank, world)
    M, N, m, n = (C.c_int64() for _ in range(4))
    _lib.check(_lib.lib().dnm_mat_sizes(h, C.byref(M), C.byref(N), C.byref(m), C.byref(n)))
    per = (1 << L) // world // (2 if flags else 1)
    assert (M.value, m.value, n.value) == ((1 << L) // (2 if flags else 1) if flags else 1 << L, per, per) or m.value == per
    d = _describe(h)
    assert "tiled=1" in d and "n_loc=%d" % (per.bit_length() - 1) in d
    sends, recvs = backend.exchange_plan(h)
    if world > 1:
        assert recvs and all(0 <= off and off + cnt <= per and cnt > 0 for _, off, cnt in sends + recvs)
        assert all(0 <= p < world and p != rank for p, _, _ in sends + recvs)
        # the transposed exchange, split inside the handle: both parts rank-local, pieces of 2^f elements
        chosen = C.c_int()
        _lib.check(_lib.lib().dnm_mat_set_exchange(h, _lib.EXCHANGE_TRANSPOSE, C.byref(chosen)))
        assert chosen.value == _lib.EXCHANGE_TRANSPOSE
        lo, hi, f = C.c_void_p(), C.c_void_p(), C.c_int()
        _lib.check(_lib.lib().dnm_mat_exchange_parts(h, C.byref(lo), C.byref(hi), C.byref(f)))
        nbits, p = per.bit_length() - 1, world.bit_length() - 1
        assert f.value == nbits - 1 - p
        for part in (lo, hi):
            assert backend.exchange_plan(part) == ([], []) and "tiled=1" in _describe(part)
        pieces, own, cnt = backend.transpose_pieces(nbits, p, f.value, rank)
        assert cnt == 1 << f.value and sum(c for _, _, c in pieces) + cnt * len(own) == per
        assert max(off + c for _, off, c in pieces) <= per
    else:
        assert sends == [] and recvs == []
    _lib.check(_lib.lib().dnm_mat_destroy(h))


def test_parity_plan_at_32_local_bits():
    L = 33
    sub = Parity('even', L=L)
    h = backend.create_mat(*_arrays(models.ising(L)), sub._c(), sub._c(), False, _lib.MAT_HOST_ONLY, 0, 1)
    assert "n_loc=32" in _describe(h)
    _lib.check(_lib.lib().dnm_mat_destroy(h))


def _colex_rank(state, L):
    """S2I of SpinConserve (bsubspace_impl.h:191-208): sum over the set bits p_1 < p_2 < ... of C(p_j, j)."""
    r, j = 0, 0
    for p in range(L):
        if (state >> p) & 1:
            j += 1
            r += math.comb(p, j)
    return r


@pytest.mark.parametrize("flags", [0, _lib.MAT_REAL_PACKED])
def test_config5_tables_of_every_rank(flags):
    L, k, P = 36, 18, 8
    dim = math.comb(L, k)
    assert dim == 9075135300 and dim > 2 ** 33
    sub = SpinConserve(L, k)
    d = _lib.Subspace.from_buffer_copy(sub._c())
    d.vec_swizzle = 14 | (10 << 8)
    arrs = _arrays(models.heisenberg(L))
    nint = C.c_int64()
    _lib.check(_lib.lib().dnm_vec_layout_size(C.byref(d), C.byref(nint)))
    assert dim <= nint.value < dim * 1.01
    at_layout, at_ref = 0, 0
    rs = np.random.RandomState(7)
    for r in range(P):
        istart, ilen, nstart, nlen = backend.layout_partition(d, P, r)
        assert (istart, nstart) == (at_layout, at_ref) and ilen >= nlen > 0
        at_layout, at_ref = istart + ilen, nstart + nlen
        assert 0.8 * dim / P < nlen < 1.2 * dim / P            # whole blocks of equal top bits, balanced within 20 %
        h = backend.create_mat(*arrs, d, d, False, _lib.MAT_HOST_ONLY | flags, r, P)
        desc = _describe(h)
        assert "internal layout" in desc
        lo, hi = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().dnm_mat_column_window(h, C.byref(lo), C.byref(hi), None))
        unit = 2 if flags else 1           # a real-packed handle counts pairs of positions
        assert 0 <= lo.value * unit <= istart and istart + ilen <= (hi.value + 1) * unit <= nint.value + unit
        shift = max(0, int(hi.value - lo.value + 1).bit_length() - 11)
        n = (hi.value >> shift) - (lo.value >> shift) + 1
        cmap = np.zeros(n, dtype=np.uint8)
        _lib.check(_lib.lib().dnm_mat_column_chunks(h, shift, cmap.ctypes.data_as(C.POINTER(C.c_uint8)), n, None))
        need = backend.needed_ranges(cmap, shift, (lo.value, hi.value))
        assert need and need[0][0] >= lo.value and need[-1][1] <= hi.value + 1
        own = (istart // unit, (istart + ilen) // unit)
        assert any(a <= own[0] and own[1] <= b for a, b in need) or sum(b - a for a, b in need) >= own[1] - own[0]
        remote = sum(max(0, min(b, own[0]) - a) + max(0, b - max(a, own[1])) for a, b in need)
        assert 0 < remote * 16 * unit < 64 * 2 ** 30           # what travels to this rank per multiply: below the 60 GiB window
        # positions of sampled indices of this rank: a bijection into its block, consistent with the colex rank
        part = _lib.Partition(r, P)
        idx = np.unique(np.concatenate([[0, nlen - 1], rs.randint(0, nlen, size=200)])).astype(np.int64)
        pos = np.empty_like(idx)
        _lib.check(_lib.lib().dnm_vec_layout_positions_host(C.byref(d), C.byref(part), idx.size, _lib.p64(idx), _lib.p64(pos)))
        assert pos.min() >= 0 and pos.max() < ilen and np.unique(pos).size == idx.size
        states = np.empty_like(idx)
        gidx = np.ascontiguousarray(idx + nstart)
        _lib.check(_lib.lib().dnm_idx_to_state(C.byref(d), gidx.size, _lib.p64(gidx), _lib.p64(states)))
        for g, s_ in zip(gidx[:40].tolist(), states[:40].tolist()):
            assert bin(s_).count("1") == k and _colex_rank(s_, L) == g
        back = np.empty_like(idx)
        _lib.check(_lib.lib().dnm_state_to_idx(C.byref(d), states.size, _lib.p64(states), _lib.p64(back)))
        assert np.array_equal(back, gidx)
        _lib.check(_lib.lib().dnm_mat_destroy(h))
    assert at_ref == dim and at_layout == nint.value


def test_kagome36_relabelling_and_handle():
    """The largest cluster one GPU solves (XParity(SpinConserve(36, 18)), 4.54 G representatives): the site relabelling is
    a permutation that keeps spin L-1, the hop counts add up to the 72 bonds, the host-only handle plans the bond-graph
    passes."""
    H = models.kagome("36a")
    masks, offs, signs, coeffs = _arrays(H)
    L = H.L
    perm, counts = backend.choose_site_perm(masks, L, 14, 10, fix_top=True)
    assert sorted(perm.tolist()) == list(range(L)) and perm[L - 1] == L - 1
    assert sum(counts) == 72 and counts[0] + counts[1] + counts[2] >= 36        # at least half the bonds inside a field
    sub = SpinConserve(L, L // 2)
    d = _lib.Subspace.from_buffer_copy(sub._c())
    d.vec_swizzle = 14 | (10 << 8)
    dp = backend.with_site_perm(d, perm)
    h = backend.create_mat(masks, offs, signs, coeffs, dp, dp, False, _lib.MAT_HOST_ONLY, 0, 1)
    assert "bond graph" in _describe(h)
    _lib.check(_lib.lib().dnm_mat_destroy(h))


def test_config5_exchange_on_the_partition_made_for_it():
    """BASELINE configs[4] on 8 ranks: what every rank receives per multiply in the reference-compatible block order
    (ascending T: ranges of the reference order) and in the order made for partitions (dnm_subspace.vec_swizzle bits
    16-19 = 1: contiguous ranges cut ONE bond of the chain), from host-only handles -- exact runs of needed blocks
    (dnm_mat_column_ranges).  The busiest rank's 42 GiB become 14."""
    L, k, P = 36, 18, 8
    sub = SpinConserve(L, k)
    arrs = _arrays(models.heisenberg(L))
    recv = {}
    for order in (0, 1):
        d = _lib.Subspace.from_buffer_copy(sub._c())
        d.vec_swizzle = 14 | (10 << 8) | (order << 16)
        nint = C.c_int64()
        _lib.check(_lib.lib().dnm_vec_layout_size(C.byref(d), C.byref(nint)))
        at, out, nranges, wins = 0, [], [], []
        for r in range(P):
            istart, ilen, nstart, nlen = backend.layout_partition(d, P, r)
            assert istart == at and ilen > 0 and 0.95 * math.comb(L, k) / P < nlen < 1.05 * math.comb(L, k) / P
            assert (nstart >= 0) == (order == 0 or r == 0)        # a reference side only in the reference-compatible order
            at += ilen
            h = backend.create_mat(*arrs, d, d, False, _lib.MAT_HOST_ONLY, r, P)
            lo, hi = C.c_int64(), C.c_int64()
            _lib.check(_lib.lib().dnm_mat_column_window(h, C.byref(lo), C.byref(hi), None))
            n = C.c_int64()
            _lib.check(_lib.lib().dnm_mat_column_ranges(h, 0, None, C.byref(n)))
            assert 0 < n.value <= 1024                              # fits the record the ranks exchange (csrc/comm.cpp)
            rg = (C.c_int64 * (2 * n.value))()
            _lib.check(_lib.lib().dnm_mat_column_ranges(h, n.value, rg, C.byref(n)))
            need = [(rg[2 * i], rg[2 * i + 1]) for i in range(n.value)]
            assert all(a < b for a, b in need) and all(need[i][1] < need[i + 1][0] for i in range(len(need) - 1))
            assert need[0][0] == lo.value and need[-1][1] == hi.value + 1
            assert any(a <= istart and istart + ilen <= b for a, b in need)       # a rank reads its own rows
            out.append(sum(max(0, min(b, istart) - a) + max(0, b - max(a, istart + ilen)) for a, b in need) * 16 / 2 ** 30)
            nranges.append(n.value)
            wins.append((hi.value - lo.value + 1) * 16 / 2 ** 30)
            _lib.check(_lib.lib().dnm_mat_destroy(h))
        assert at == nint.value
        recv[order] = out
        assert max(wins) < 90                                       # GiB of window per rank (complex128; half in real arithmetic)
    assert 42 < max(recv[0]) < 43 and 25 < sum(recv[0]) / P < 26
    assert max(recv[1]) < 14.5 and sum(recv[1]) / P < 11.5
    assert max(recv[1]) < 0.35 * max(recv[0])
