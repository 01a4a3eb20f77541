"""
world_size-2 (and 4) test of the partitioned multiply on CPU with the gloo
backend: every rank builds its own plan (real host code, DNM_MAT_HOST_ONLY),
exchanges blocks with its XOR partners through torch.distributed exactly as
ShellMat.mult does, and applies the rank-local and partner passes through the
kernel emulation.  The gathered result must equal the oracle's multi-rank
MatMult_CPU_Fast.  Also covers the small all-reduces the Krylov hooks use.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, L, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DNM_TILE_BITS="8", DNM_LOG_ROWS="2",
                      DNM_PLAN_MODE="2", DNM_GBITS="3")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamite_amd import models, msc_tools
    from dynamite_amd.subspaces import Full
    from plan_emulator import HostMat, run_pass, run_remote
    from dynamite_amd.backend import post_exchange

    H = models.mbl(L)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    sub = Full(L=L)
    hm = HostMat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._c(), sub._c(), rank=rank, nranks=world)
    nloc = (1 << L) // world
    rs = np.random.RandomState(11)          # same global vector on every rank, each keeps its block
    xg = rs.standard_normal(1 << L) + 1j * rs.standard_normal(1 << L)
    # the block as it sits in device memory: the subspace's vector layout on the local index (with four ranks the
    # shift is the smaller one the transposed exchange asks for, subspaces.Subspace.vec_swizzle)
    from plan_emulator import vec_pos
    pos = vec_pos(np.arange(nloc), sub.vec_swizzle)
    xl = np.empty(nloc, dtype=complex)
    xl[pos] = xg[rank * nloc:(rank + 1) * nloc]
    x = torch.from_numpy(xl)

    # the exchange of ShellMat.mult: post all sends/recvs, do local work, wait, partner passes
    bufs = [torch.empty(cnt, dtype=x.dtype) for _, _, cnt in hm.recvs]
    reqs = post_exchange(x, hm.sends, hm.recvs, bufs)
    y = np.zeros(nloc, dtype=complex)
    for ps in hm.local:
        run_pass(hm, ps, x.numpy(), y)
    for r in reqs:
        r.wait()
    for i in range(len(hm.recvs)):
        run_remote(hm, i, bufs[i].numpy(), y)
    sent = sum(c for _, _, c in hm.sends)
    y = y[pos]                                  # back to index order
    x = torch.from_numpy(xg[rank * nloc:(rank + 1) * nloc].copy())

    # Krylov-style reductions: global <x|y> and max |y|
    t = torch.tensor([np.vdot(x.numpy(), y).real, np.vdot(x.numpy(), y).imag], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    m = torch.tensor([np.abs(y).max()], dtype=torch.float64)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    parts = [torch.empty(nloc, dtype=torch.complex128) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(y))
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), y=torch.cat(parts).numpy(), x=xg, dot=t.numpy(),
                 mx=m.numpy(), partners=np.array(hm.partners), sent=sent)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_partitioned_multiply_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    from dynamite_amd import models, msc_tools
    L = 14
    port = _free_port()
    mp.spawn(_worker, args=(world, port, L, str(tmp_path)), nprocs=world, join=True)
    res = np.load(tmp_path / "result.npz")
    H = models.mbl(L)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    msc = orc.Msc(masks, offs, H.msc['signs'], H.msc['coeffs'])
    ref = orc.matvec_fast_ranks(msc, orc.full(L), res["x"], world)
    assert np.max(np.abs(res["y"] - ref)) < 30 * 64 * 2.2e-16 * np.abs(res["x"]).max()
    d = np.vdot(res["x"], ref)
    assert abs(complex(res["dot"][0], res["dot"][1]) - d) < 1e-9
    assert abs(res["mx"][0] - np.abs(ref).max()) < 1e-12
    assert len(res["partners"]) == 1
    # rank 0 (all rank bits 0): the flip-flop over the block boundary needs half a block from rank 1,
    # the bond inside the rank bits (world 4) annihilates its rows
    assert int(res["sent"]) == (1 << L) // world // 2


def _window_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamite_amd.backend import split_ownership, exchange_window
    N = 1003
    xg = torch.arange(N, dtype=torch.float64).to(torch.complex128) * (1 + 2j)
    owned = [split_ownership(N, world, q) for q in range(world)]
    # windows reach unevenly into the neighbours (and, for rank 0, across two ranks)
    windows = []
    for q, (s, n) in enumerate(owned):
        lo = max(0, s - 17 * (q + 1))
        hi = min(N - 1, s + n - 1 + (400 if q == 0 else 29))
        windows.append((lo, hi))
    s, n = owned[rank]
    buf = exchange_window(xg[s:s + n].clone(), owned, windows, rank)
    lo, hi = windows[rank]
    ok = torch.equal(buf, xg[lo:hi + 1])
    # only the ranges a rank reads (backend.needed_ranges of the device's chunk map): holes stay as allocated
    needs = []
    for q, (wlo, whi) in enumerate(windows):
        third = (whi + 1 - wlo) // 3
        needs.append([(wlo, wlo + third - 5 * q), (whi + 1 - third, whi + 1)])
    buf2 = exchange_window(xg[s:s + n].clone(), owned, windows, rank, None, needs)
    want = torch.zeros_like(buf2)
    for a, b in needs[rank] + [(max(lo, s), min(hi + 1, s + n))]:
        want[a - lo:b - lo] = xg[a:b]
    ok = ok and torch.equal(buf2, want)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(os.path.join(out_dir, "ok.txt"), "w").write(str(flag.item()))
    dist.destroy_process_group()


def test_window_exchange_gloo(tmp_path):
    """The column-window exchange of the partitioned SpinConserve multiply (uneven
    blocks, windows spanning several ranks) on 3 gloo ranks."""
    import torch.multiprocessing as mp
    mp.spawn(_window_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert float(open(tmp_path / "ok.txt").read()) == 1.0


def _redistribute_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamite_amd import backend
    from dynamite_amd.subspaces import SpinConserve
    ok = True
    # the two partitions a SpinConserve vector in the internal layout and its reference-order copy have on several
    # ranks: whole top-bit blocks (dnm_vec_layout_partition) against PetscSplitOwnership -- L=14, k=6, (a, w) = (6, 4)
    sub = SpinConserve(14, 6)
    d = sub._c()
    d.vec_swizzle = 6 | (4 << 8)
    n = sub.get_dimension()
    lay = [backend.layout_partition(d, world, q)[2:4] for q in range(world)]
    ref = [backend.split_ownership(n, world, q) for q in range(world)]
    assert sum(c for _, c in lay) == n and lay != ref
    xg = torch.arange(n, dtype=torch.float64).to(torch.complex128) * (1 - 3j)
    s, c = lay[rank]
    there = backend.redistribute(xg[s:s + c].clone(), lay, ref, rank)
    s2, c2 = ref[rank]
    ok = ok and torch.equal(there, xg[s2:s2 + c2])
    back = backend.redistribute(there, ref, lay, rank)
    ok = ok and torch.equal(back, xg[s:s + c])
    # a partition with an empty rank and one with everything on one rank
    odd = [(0, 0), (0, n - 5), (n - 5, 5)][:world] if world == 3 else [(0, n)] + [(n, 0)] * (world - 1)
    s3, c3 = odd[rank]
    got = backend.redistribute(xg[s2:s2 + c2].clone(), ref, odd, rank)
    ok = ok and torch.equal(got, xg[s3:s3 + c3])
    try:
        backend.redistribute(xg[:3], lay, ref, rank)
        ok = False
    except ValueError:
        pass
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(os.path.join(out_dir, "ok_redist.txt"), "w").write(str(flag.item()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_redistribute_between_partitions_gloo(tmp_path, world):
    """backend.redistribute: a vector moves between the internal layout's partition and the reference-order one
    (what ShellMat._mult_converted does on several ranks), and through partitions with empty ranks."""
    import torch.multiprocessing as mp
    mp.spawn(_redistribute_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert float(open(tmp_path / "ok_redist.txt").read()) == 1.0


def _reorder_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes as C
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamite_amd import backend, _lib
    from dynamite_amd.subspaces import SpinConserve
    ok = True
    for (L, k, a, w) in ((14, 6, 6, 4), (16, 8, 6, 4), (15, 4, 8, 2)):
        sub = SpinConserve(L, k)
        n = sub.get_dimension()
        ds, glob = [], []
        for order in (0, 1):
            d = sub._c()
            d.vec_swizzle = a | (w << 8) | (order << 16)
            nint = C.c_int64()
            _lib.check(_lib.lib().dnm_vec_layout_size(C.byref(d), C.byref(nint)))
            idx = np.arange(n, dtype=np.int64)
            pos = np.empty_like(idx)
            _lib.check(_lib.lib().dnm_vec_layout_positions_host(C.byref(d), None, n, _lib.p64(idx), _lib.p64(pos)))
            g = np.zeros(nint.value, dtype=np.complex128)
            g[pos] = (idx + 1) * (1 - 3j)                      # keyed by reference index; padding stays zero
            ds.append(d)
            glob.append(torch.from_numpy(g))
        shares = [backend.layout_partition(d, world, rank)[:2] for d in ds]
        mine = [g[s:s + c].clone() for g, (s, c) in zip(glob, shares)]
        T0, ib0 = backend.layout_blocks(ds[0])
        T1, ib1 = backend.layout_blocks(ds[1])
        ok = ok and sorted(T0) == sorted(T1) and list(T0) == sorted(T0) and ib0[-1] == ib1[-1] == glob[0].numel()
        there = backend.reorder_blocks(mine[0], ds[0], ds[1])
        ok = ok and torch.equal(there, mine[1])
        back = backend.reorder_blocks(there, ds[1], ds[0])
        ok = ok and torch.equal(back, mine[0])
        # real vectors stored two positions to an element: the same moves on half as many elements
        if all(c % 2 == 0 for _, c in shares):
            rp = [torch.view_as_complex(m.real.contiguous().reshape(-1, 2)) for m in mine]
            got = backend.reorder_blocks(rp[0], ds[0], ds[1], per_position=1)
            ok = ok and torch.equal(got, rp[1])
        try:
            backend.reorder_blocks(mine[0][:-1], ds[0], ds[1])
            ok = False
        except ValueError:
            pass
    # layouts that differ in more than the block order are refused
    d2 = sub._c()
    d2.vec_swizzle = 7 | (2 << 8)
    try:
        backend.block_moves(ds[0], d2, world)
        ok = False
    except (ValueError, _lib.BackendError):
        pass
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(os.path.join(out_dir, "ok_reorder.txt"), "w").write(str(flag.item()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_reorder_blocks_between_block_orders_gloo(tmp_path, world):
    """backend.reorder_blocks: a rank's share of a SpinConserve vector moves between the reference-compatible block order
    and the one made for partitions (whole T blocks over a ring of sends and receives) and back, for complex vectors and
    for real ones stored in pairs; what a vector holds at every reference index is the same in both."""
    import torch.multiprocessing as mp
    mp.spawn(_reorder_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert float(open(tmp_path / "ok_reorder.txt").read()) == 1.0


def _layout_choice_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes as C
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamite_amd import backend, models, msc_tools, _lib
    from dynamite_amd.config import config
    from dynamite_amd.subspaces import SpinConserve
    ok = config.world_size == world
    # L=26, k=13 under the production configuration on four ranks: 4 blocks of equal top bits cannot be shared out
    #  -> reference order, PETSc's even split, the window kernels
    sub = SpinConserve(26, 13)
    ok = ok and sub.vec_swizzle == 0
    H = models.heisenberg(26)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    h = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._c(), sub._c(), False, _lib.MAT_HOST_ONLY,
                           rank, world)
    M, N, m, n = (C.c_int64() for _ in range(4))
    _lib.check(_lib.lib().dnm_mat_sizes(h, C.byref(M), C.byref(N), C.byref(m), C.byref(n)))
    ok = ok and (rank * (M.value // world) + min(rank, M.value % world), m.value) == backend.split_ownership(M.value, world, rank)
    ll, lr = C.c_int(), C.c_int()
    _lib.check(_lib.lib().dnm_mat_layouts(h, C.byref(ll), C.byref(lr)))
    ok = ok and (ll.value, lr.value) == (0, 0)
    _lib.check(_lib.lib().dnm_mat_destroy(h))
    # ... while a subspace whose blocks balance keeps the internal layout on the same four ranks
    ok = ok and SpinConserve(32, 16).vec_swizzle == (14 | (10 << 8))
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(os.path.join(out_dir, "ok_layout.txt"), "w").write(str(flag.item()))
    dist.destroy_process_group()


def test_spinconserve_layout_choice_on_four_ranks(tmp_path):
    """SpinConserve.vec_swizzle under a real process group: L=26, k=13 on 4 ranks stays in reference order (every rank
    owns a quarter of the rows), L=32, k=16 takes the internal layout."""
    import torch.multiprocessing as mp
    mp.spawn(_layout_choice_worker, args=(4, _free_port(), str(tmp_path)), nprocs=4, join=True)
    assert float(open(tmp_path / "ok_layout.txt").read()) == 1.0
