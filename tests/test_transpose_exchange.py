"""
The transposed exchange of the partitioned Full-space multiply (backend.transpose_split / transpose_pieces /
ShellMat._mult_transposed) on CPU: the split and the bit permutation against the oracle on one process, the
piece list against the index map it has to realise, and the whole multiply on 2 and 4 gloo ranks with the host
plans run through the kernel emulation, against the oracle's multi-rank MatMult_CPU_Fast
(reference: bpetsc_template_2.c:713-889; what it replaces: the VecScatter of :787-879).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _arrays(name, L):
    from dynamite_amd import models, msc_tools
    H = models.BY_NAME[name](L)
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    return masks, offs, H.msc['signs'], H.msc['coeffs']


def _swap_index(g, n, p, f):
    d = ((g >> f) ^ (g >> n)) & ((1 << p) - 1)
    return g ^ (d << f) ^ (d << n)


@pytest.mark.parametrize("name,L,P", [("mbl", 10, 4), ("heisenberg", 11, 8), ("ising", 9, 2), ("localized", 10, 4),
                                      ("long_range", 10, 4)])
def test_split_reproduces_the_operator(name, L, P):
    """H x = H_lo x + S (H_hi' (S x)) with S the swap of the rank bits and the field F."""
    from oracle import oracle as orc
    from dynamite_amd.backend import transpose_split
    arrs = _arrays(name, L)
    split = transpose_split(*arrs, L, P)
    assert split is not None
    lo, hi, f = split
    p = P.bit_length() - 1
    n = L - p
    assert f == n - 1 - p
    # the two parts hold every term once; the permuted part flips no spin of the new rank field
    assert lo[2].size + hi[2].size == arrs[2].size
    assert not (lo[0] >> n).any() and (hi[0] >> n).any() is not None
    assert not ((hi[0] >> n) & (P - 1)).any()
    rs = np.random.RandomState(5)
    x = rs.standard_normal(1 << L) + 1j * rs.standard_normal(1 << L)
    sub = orc.full(L)
    ref = orc.matvec(orc.Msc(*arrs), sub, sub, x)
    perm = _swap_index(np.arange(1 << L, dtype=np.int64), n, p, f)
    y = orc.matvec(orc.Msc(*lo), sub, sub, x)
    wb = orc.matvec(orc.Msc(*hi), sub, sub, x[perm])          # layout B holds x[swap(g')] at g'
    y = y + wb[perm]
    assert np.max(np.abs(y - ref)) <= 1e-13 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name,L,P,space", [("mbl", 11, 4, 0), ("heisenberg", 12, 8, 1), ("long_range", 10, 2, 1)])
def test_split_reproduces_the_operator_parity(name, L, P, space):
    """The same on a Parity subspace: index bit j is spin j + 1, the fields move as spins."""
    from oracle import oracle as orc
    from dynamite_amd.backend import transpose_split
    arrs = _arrays(name, L)
    nb = L - 1
    split = transpose_split(*arrs, nb, P, shift=1)
    assert split is not None
    lo, hi, f = split
    p = P.bit_length() - 1
    n = nb - p
    rs = np.random.RandomState(6)
    x = rs.standard_normal(1 << nb) + 1j * rs.standard_normal(1 << nb)
    sub = orc.parity(L, space)
    ref = orc.matvec(orc.Msc(*arrs), sub, sub, x)
    perm = _swap_index(np.arange(1 << nb, dtype=np.int64), n, p, f)
    y = orc.matvec(orc.Msc(*lo), sub, sub, x)
    wb = orc.matvec(orc.Msc(*hi), sub, sub, x[perm])
    y = y + wb[perm]
    assert np.max(np.abs(y - ref)) <= 1e-13 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name,L,P,kind", [("mbl", 16, 4, "full"), ("heisenberg", 17, 8, "full"), ("ising", 15, 2, "full"),
                                           ("long_range", 16, 4, "full"), ("mbl", 17, 4, "parity"),
                                           ("heisenberg", 17, 8, "packed")])
def test_native_split_is_the_host_split(name, L, P, kind):
    """dnm_mat_set_exchange (csrc/mat.cpp) splits inside the handle what backend.transpose_split splits on the host: the
    two parts of every rank, term by term (dnm_mat_operator), the field, and the plans the parts get (host-only handles:
    no device needed)."""
    import ctypes as C
    from dynamite_amd import _lib, backend
    from dynamite_amd.subspaces import Full, Parity
    arrs = _arrays(name, L)
    sub = Parity('even', L=L) if kind == "parity" else Full(L=L)
    shift = 1 if kind == "parity" else 0
    flags = _lib.MAT_HOST_ONLY | (_lib.MAT_REAL_PACKED if kind == "packed" else 0)
    sc = sub._to_c()['data']
    sc.vec_swizzle = 6 if name == "heisenberg" else 0              # (pieces of 2^f >= 2^8 amplitudes: a swizzle of 6 fits)
    split = backend.transpose_split(*arrs, L - shift, P, int(sc.vec_swizzle), shift, packed=kind == "packed")
    assert split is not None
    lib = _lib.lib()

    def operator_of(h):
        nm, nt = C.c_int64(), C.c_int64()
        _lib.check(lib.dnm_mat_operator(h, C.byref(nm), C.byref(nt), None, None, None, None))
        m, o = np.zeros(nm.value, np.int64), np.zeros(nm.value + 1, np.int64)
        s_, c = np.zeros(nt.value, np.int64), np.zeros(nt.value, np.float64)
        _lib.check(lib.dnm_mat_operator(h, C.byref(nm), C.byref(nt), _lib.p64(m), _lib.p64(o), _lib.p64(s_), _lib.pf64(c)))
        return m, o, s_, c

    def describe(h):
        buf = C.create_string_buffer(1 << 14)
        _lib.check(lib.dnm_mat_plan_describe(h, buf, len(buf)))
        return buf.value.decode()

    for rank in (0, P - 1, P // 2):
        h = backend.create_mat(*arrs, sc, sc, False, flags, rank, P)
        chosen = C.c_int()
        _lib.check(lib.dnm_mat_set_exchange(h, _lib.EXCHANGE_AUTO, C.byref(chosen)))
        assert chosen.value == (_lib.EXCHANGE_TRANSPOSE if P >= 4 else _lib.EXCHANGE_PARTNER)
        _lib.check(lib.dnm_mat_set_exchange(h, _lib.EXCHANGE_TRANSPOSE, C.byref(chosen)))
        assert chosen.value == _lib.EXCHANGE_TRANSPOSE
        lo, hi, f = C.c_void_p(), C.c_void_p(), C.c_int()
        _lib.check(lib.dnm_mat_exchange_parts(h, C.byref(lo), C.byref(hi), C.byref(f)))
        assert f.value == split[2] - (1 if kind == "packed" else 0)
        for part, want in ((lo, split[0]), (hi, split[1])):
            m, o, s_, c = operator_of(part)
            assert np.array_equal(m, want[0]) and np.array_equal(o, want[1]) and np.array_equal(s_, want[2])
            wc = np.where(want[3].real != 0, want[3].real, want[3].imag)
            assert np.array_equal(c, wc)
        n = (L - shift - (P.bit_length() - 1)) - (1 if kind == "packed" else 0)
        fl = flags
        a = 12 - (n - f.value)
        if n - f.value <= 8 and 2 <= a <= 9:
            fl |= a << _lib.MAT_AMIN_SHIFT
        ref_hi = backend.create_mat(*split[1], sc, sc, False, fl, rank, P)
        assert describe(hi) == describe(ref_hi)
        lib.dnm_mat_destroy(ref_hi)
        _lib.check(lib.dnm_mat_set_exchange(h, _lib.EXCHANGE_PARTNER, C.byref(chosen)))
        _lib.check(lib.dnm_mat_exchange_parts(h, C.byref(lo), C.byref(hi), C.byref(f)))
        assert chosen.value == _lib.EXCHANGE_PARTNER and not lo.value and not hi.value
        lib.dnm_mat_destroy(h)


@pytest.mark.parametrize("L,P", [(17, 4), (18, 8)])
def test_shellmat_reports_the_native_transposed_exchange_like_the_host_one(L, P):
    """backend.ShellMat with the split inside the library (set_native_transposed) tells the same story as with the host's
    split (set_transposed): scheme, bytes per multiply, peers, busiest link, kernel launches (host-only handles)."""
    from dynamite_amd import _lib, backend
    from dynamite_amd.subspaces import Full
    arrs = _arrays("mbl", L)
    sc = Full(L=L)._to_c()['data']
    sc.vec_swizzle = 0
    for rank in (0, P - 1):
        h = backend.create_mat(*arrs, sc, sc, False, _lib.MAT_HOST_ONLY, rank, P)
        m = backend.ShellMat(h, sc, sc, P, rank)
        assert m.set_native_transposed()
        h2 = backend.create_mat(*arrs, sc, sc, False, _lib.MAT_HOST_ONLY, rank, P)
        m2 = backend.ShellMat(h2, sc, sc, P, rank)
        m2.set_transposed(backend.transpose_split(*arrs, L, P, 0), sc, sc, _lib.MAT_HOST_ONLY)
        assert m.exchange_summary() == m2.exchange_summary() and m.exchange_summary()["scheme"] == "transpose"
        assert m2._tr_pipe and m.launches_per_mult() == m2.launches_per_mult()
        m.destroy()
        m2.destroy()


def test_native_split_refuses_what_the_host_split_refuses():
    import ctypes as C
    from dynamite_amd import _lib, backend, msc_tools
    from dynamite_amd.operators import sigmax, index_sum
    from dynamite_amd.subspaces import Full
    lib = _lib.lib()
    H = index_sum(sigmax(0) * sigmax(3), size=16)       # flips three sites apart couple the rank bits to the field F
    H.L = 16
    H.reduce_msc()
    m, o = msc_tools.get_mask_offsets(H.msc)
    sc = Full(L=16)._to_c()['data']
    for arrs, P in (((m, o, H.msc['signs'], H.msc['coeffs']), 4), (_arrays("mbl", 16), 1)):
        h = backend.create_mat(*arrs, sc, sc, False, _lib.MAT_HOST_ONLY, 0, P)
        chosen = C.c_int(7)
        _lib.check(lib.dnm_mat_set_exchange(h, _lib.EXCHANGE_TRANSPOSE, C.byref(chosen)))
        assert chosen.value == _lib.EXCHANGE_PARTNER
        assert lib.dnm_mat_set_exchange(h, 5, C.byref(chosen)) != 0
        lib.dnm_mat_destroy(h)


def test_split_refuses_what_it_cannot_do():
    from dynamite_amd.backend import transpose_split
    from dynamite_amd import msc_tools
    from dynamite_amd.operators import sigmax, index_sum
    # flips three sites apart couple the top spins to the field F
    H = index_sum(sigmax(0) * sigmax(3), size=10)
    H.L = 10
    H.reduce_msc()
    m, o = msc_tools.get_mask_offsets(H.msc)
    assert transpose_split(m, o, H.msc['signs'], H.msc['coeffs'], 10, 4) is None
    # nothing crosses the ranks
    H = index_sum(sigmax(0), size=6)
    H.L = 8
    H.reduce_msc()
    m, o = msc_tools.get_mask_offsets(H.msc)
    assert transpose_split(m, o, H.msc['signs'], H.msc['coeffs'], 8, 4) is None
    # pieces must lie above the swizzle field; three ranks are not a power of two; too few local spins
    arrs = _arrays("mbl", 12)
    assert transpose_split(*arrs, 12, 4, swizzle=6) is None        # f = 7 < 8
    assert transpose_split(*arrs, 12, 4, swizzle=5) is not None    # f = 7 >= 6
    assert transpose_split(*arrs, 12, 3) is None
    assert transpose_split(*_arrays("mbl", 5), 5, 8) is None


@pytest.mark.parametrize("L,P", [(10, 4), (12, 8), (9, 2)])
def test_pieces_realise_the_swap(L, P):
    from dynamite_amd.backend import transpose_pieces
    p = P.bit_length() - 1
    n = L - p
    f = n - 1 - p
    g = np.arange(1 << L, dtype=np.int64)
    blocks = [g[r << n:(r + 1) << n].copy() for r in range(P)]
    out = [np.full(1 << n, -1, dtype=np.int64) for _ in range(P)]
    for r in range(P):
        pieces, own, cnt = transpose_pieces(n, p, f, r)
        assert len(pieces) == (P - 1) * (1 << (n - f - p)) and cnt == 1 << f
        for off in own:
            out[r][off:off + cnt] = blocks[r][off:off + cnt]
        for q, off, c in pieces:           # the piece at `off` goes to q and lands there at r's offset
            dst = [o for (qq, o, _) in transpose_pieces(n, p, f, q)[0] if qq == r]
            src = [o for (qq, o, _) in pieces if qq == q]
            out[q][dst[src.index(off)]:dst[src.index(off)] + c] = blocks[r][off:off + c]
    layout_b = np.concatenate(out)
    assert np.array_equal(layout_b, _swap_index(g, n, p, f))      # position g' holds element swap(g')


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, L, name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DNM_TILE_BITS="8", DNM_LOG_ROWS="2",
                      DNM_PLAN_MODE="2", DNM_GBITS="3", DNM_EXCHANGE="transpose", DNM_SWZ="6")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamite_amd.subspaces import Full
    from dynamite_amd.backend import transpose_split, transpose_pieces, post_transpose
    from plan_emulator import HostMat, run_pass

    arrs = _arrays(name, L)
    sub = Full(L=L)
    lo, hi, f = transpose_split(*arrs, L, world, sub.vec_swizzle)
    sc = sub._c()
    mats = [HostMat(*part, sc, sc, rank=rank, nranks=world) for part in (lo, hi)]
    assert all(not m.recvs and not m.sends and m.tiled for m in mats)
    p = world.bit_length() - 1
    n = L - p
    nloc = 1 << n
    pieces, own, cnt = transpose_pieces(n, p, f, rank)
    rs = np.random.RandomState(11)
    xg = rs.standard_normal(1 << L) + 1j * rs.standard_normal(1 << L)
    # the block as it sits in device memory (swizzled on the local index; pieces lie above the field)
    from plan_emulator import vec_pos
    S = sub.vec_swizzle
    assert S == 0 or f >= 2 * S - 4
    pos = vec_pos(np.arange(nloc), S)
    xl = np.empty(nloc, dtype=complex)
    xl[pos] = xg[rank * nloc:(rank + 1) * nloc]
    x = torch.from_numpy(xl)

    # ShellMat._mult_transposed, with the passes run by the emulation
    xb, wb = torch.empty_like(x), torch.zeros_like(x)
    reqs = post_transpose(x, xb, pieces)
    for off in own:
        xb[off:off + cnt] = x[off:off + cnt]
    y = np.zeros(nloc, dtype=complex)
    for ps in mats[0].local:
        run_pass(mats[0], ps, x.numpy(), y)
    for r in reqs:
        r.wait()
    w = np.zeros(nloc, dtype=complex)
    for ps in mats[1].local:
        run_pass(mats[1], ps, xb.numpy(), w)
    wb.copy_(torch.from_numpy(w))
    reqs = post_transpose(wb, xb, pieces)
    for off in own:
        y[off:off + cnt] += w[off:off + cnt]
    for r in reqs:
        r.wait()
    edges = [0] + [e for off in own for e in (off, off + cnt)] + [nloc]
    for a, b in zip(edges[0::2], edges[1::2]):
        y[a:b] += xb.numpy()[a:b]
    parts = [torch.empty(nloc, dtype=torch.complex128) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(y[pos]))
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), y=torch.cat(parts).numpy(), x=xg,
                 passes=np.array([len(m.local) for m in mats]), swz=S)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(4, "mbl"), (2, "mbl"), (4, "ising")])
def test_transposed_multiply_gloo(tmp_path, world, name):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    L = 14
    mp.spawn(_worker, args=(world, _free_port(), L, name, str(tmp_path)), nprocs=world, join=True)
    res = np.load(tmp_path / "result.npz")
    arrs = _arrays(name, L)
    ref = orc.matvec_fast_ranks(orc.Msc(*arrs), orc.full(L), res["x"], world)
    assert np.max(np.abs(res["y"] - ref)) < 30 * 64 * 2.2e-16 * np.abs(res["x"]).max()


def _worker_packed(rank, world, port, L, name, out_dir):
    """The transposed exchange of a REAL-PACKED operator (DNM_MAT_REAL_PACKED: index bit 0 is the lane of an element,
    the exchanged field and the pieces sit one bit lower): ShellMat._mult_transposed on packed vectors, the two layout
    operators built packed, passes run by the emulation."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DNM_TILE_BITS="8", DNM_LOG_ROWS="2",
                      DNM_PLAN_MODE="2", DNM_GBITS="3", DNM_EXCHANGE="transpose", DNM_SWZ="6")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamite_amd import _lib
    from dynamite_amd.subspaces import Full
    from dynamite_amd.backend import transpose_split, transpose_pieces, post_transpose
    from plan_emulator import HostMat, run_pass, vec_pos

    arrs = _arrays(name, L)
    sub = Full(L=L)
    S = sub.vec_swizzle
    lo, hi, f = transpose_split(*arrs, L, world, S, packed=True)
    sc = sub._c()
    mats = [HostMat(*part, sc, sc, rank=rank, nranks=world, flags=_lib.MAT_REAL_PACKED) for part in (lo, hi)]
    assert all(not m.recvs and not m.sends and m.tiled for m in mats)
    p = world.bit_length() - 1
    n = L - p - 1                    # local index bits of the packed operator
    fp = f - 1
    nloc = 1 << n
    assert all((1 << m.n_loc) == nloc for m in mats) and (S == 0 or fp >= 2 * S - 4)
    pieces, own, cnt = transpose_pieces(n, p, fp, rank)
    rs = np.random.RandomState(12)
    xg = rs.standard_normal(1 << L)
    xpk = xg[0::2] + 1j * xg[1::2]
    pos = vec_pos(np.arange(nloc), S)
    xl = np.empty(nloc, dtype=complex)
    xl[pos] = xpk[rank * nloc:(rank + 1) * nloc]
    x = torch.from_numpy(xl)
    xb, wb = torch.empty_like(x), torch.zeros_like(x)
    reqs = post_transpose(x, xb, pieces)
    for off in own:
        xb[off:off + cnt] = x[off:off + cnt]
    y = np.zeros(nloc, dtype=complex)
    for ps in mats[0].local:
        run_pass(mats[0], ps, x.numpy(), y)
    for r in reqs:
        r.wait()
    w = np.zeros(nloc, dtype=complex)
    for ps in mats[1].local:
        run_pass(mats[1], ps, xb.numpy(), w)
    wb.copy_(torch.from_numpy(w))
    reqs = post_transpose(wb, xb, pieces)
    for off in own:
        y[off:off + cnt] += w[off:off + cnt]
    for r in reqs:
        r.wait()
    edges = [0] + [e for off in own for e in (off, off + cnt)] + [nloc]
    for a, b in zip(edges[0::2], edges[1::2]):
        y[a:b] += xb.numpy()[a:b]
    parts = [torch.empty(nloc, dtype=torch.complex128) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(y[pos]))
    if rank == 0:
        yp = torch.cat(parts).numpy()
        yr = np.empty(1 << L)
        yr[0::2], yr[1::2] = yp.real, yp.imag
        np.savez(os.path.join(out_dir, "result_packed.npz"), y=yr, x=xg)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(4, "mbl"), (2, "heisenberg")])
def test_transposed_multiply_packed_gloo(tmp_path, world, name):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    L = 14
    mp.spawn(_worker_packed, args=(world, _free_port(), L, name, str(tmp_path)), nprocs=world, join=True)
    res = np.load(tmp_path / "result_packed.npz")
    arrs = _arrays(name, L)
    ref = orc.matvec(orc.Msc(*arrs), orc.full(L), orc.full(L), res["x"].astype(complex)).real
    assert np.max(np.abs(res["y"] - ref)) < 30 * 64 * 2.2e-16 * np.abs(res["x"]).max()
