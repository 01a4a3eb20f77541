"""
Child process of tests/test_gpu_distributed.py::test_rccl_transport_world_one: the production transport (backend
"nccl" = RCCL, device tensors handed straight to the collective -- the unstaged branch of dynamite_amd/_comm.py) on a
ONE-GPU box: a process group of world size 1 whose only peer is the rank itself.  What runs over RCCL here is what
replaces the all-gather of bcuda_template_2.cu:161-171 and the scatters of bpetsc_template_2.c:787-879:
  A  the collectives (all_reduce sum / max of the Krylov hooks, all_gather, reduce, barrier),
  B  batch_p2p with the rank as its own peer (isend + irecv of complex128 device slices in one batch),
  C  post_exchange / post_transpose with the message lists of real plans (rank 0 of 2 partner blocks, rank 1 of 4
     transposed exchange), every peer rewritten to this rank,
  D  evolve and eigsolve with the solver hooks reducing through RCCL, and exchange_only on a world-size-1 operator,
  F  the same through the native call (dnm_mat_mult_partitioned on the library's own communicator, ShellMat._native),
  E  the partitioned multiply itself (ShellMat.mult, partner blocks): rank 0 of 2 and rank 5 of 8 of a Full-space
     operator, the state chosen with all rank blocks equal so that what a partner would send is a slice of this rank's
     own block -- the production code posts the exchange on RCCL's stream, runs the rank-local passes under it, waits
     per block and applies the partner passes; the rank's rows are compared with the oracle.
Prints one JSON line: {"stage": last stage completed, "refused": RCCL's message if it refused the self send}.
Exit code 0: everything ran; 77: RCCL refused stage B (stages A and D still ran); anything else: a failure.
Started as a FRESH process by the test (never a re-exec of a process that has touched the GPU).
"""
import datetime
import faulthandler
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    faulthandler.dump_traceback_later(int(os.environ.get("DNM_TEST_HANG_S", "500")), exit=True)
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=600))
    from dynamite_amd import _comm, backend, computations, config, models, msc_tools, _lib
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import Full
    report = {"stage": "", "refused": None, "backend": dist.get_backend()}
    assert dist.get_backend() == "nccl"

    # ---- A: collectives on device memory
    rs = np.random.RandomState(5)
    t = torch.tensor(rs.standard_normal(7), dtype=torch.float64, device=dev)
    keep = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    assert torch.equal(t, keep)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert torch.equal(t, keep)
    z = torch.tensor(rs.standard_normal(1 << 12) + 1j * rs.standard_normal(1 << 12), dtype=torch.complex128, device=dev)
    assert not _comm._staged(z)
    parts = _comm.all_gather(z)
    assert len(parts) == 1 and torch.equal(parts[0], z)
    zz = z.clone()
    _comm.reduce_sum(zz)
    assert torch.equal(zz, z)
    got = _comm.gather_varied(z, [z.numel()])
    assert torch.equal(got[0], z)
    _comm.barrier()
    torch.cuda.synchronize()
    report["stage"] = "A"

    # ---- B: the rank as its own peer, slices of one allocation at non-zero offsets, two messages matched in order
    try:
        src = torch.tensor(rs.standard_normal(1 << 16) + 1j * rs.standard_normal(1 << 16), dtype=torch.complex128, device=dev)
        dst = torch.zeros(1 << 16, dtype=torch.complex128, device=dev)
        sends = [(src[4096:4096 + 8192], 0), (src[32768:32768 + 1024], 0)]
        recvs = [(dst[0:8192], 0), (dst[16384:16384 + 1024], 0)]
        for r in _comm.batch_p2p(sends, recvs):
            r.wait()
        torch.cuda.synchronize()
        assert torch.equal(dst[0:8192], src[4096:4096 + 8192]), "self send/recv, message 1"
        assert torch.equal(dst[16384:16384 + 1024], src[32768:32768 + 1024]), "self send/recv, message 2"
        assert bool((dst[8192:16384] == 0).all())
        report["stage"] = "B"

        # ---- C: message lists of real plans with every peer rewritten to this rank
        L = 20
        H = models.mbl(L)
        H.establish_L()
        H.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        c = Full(L=L)._c()
        h = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], c, c, False, _lib.MAT_HOST_ONLY, 0, 2)
        sends, recvs = backend.exchange_plan(h)
        assert sends and recvs and [s[2] for s in sends] == [r[2] for r in recvs]
        n_loc = 1 << (L - 1)
        x = torch.tensor(rs.standard_normal(n_loc) + 1j * rs.standard_normal(n_loc), dtype=torch.complex128, device=dev)
        bufs = [torch.zeros(cnt, dtype=torch.complex128, device=dev) for _, _, cnt in recvs]
        for r in backend.post_exchange(x, [(0, off, cnt) for _, off, cnt in sends], [(0, off, cnt) for _, off, cnt in recvs], bufs):
            r.wait()
        torch.cuda.synchronize()
        for (_, off, cnt), b in zip(sends, bufs):
            assert torch.equal(b, x[off:off + cnt]), "partner block through RCCL"
        _lib.check(_lib.lib().dnm_mat_destroy(h))
        pieces, own, cnt = backend.transpose_pieces(L - 2, 2, L - 2 - 1 - 2, 1)
        n4 = 1 << (L - 2)
        xa = x[:n4].clone()
        xb = torch.zeros(n4, dtype=torch.complex128, device=dev)
        for r in backend.post_transpose(xa, xb, [(0, off, c_) for _, off, c_ in pieces]):
            r.wait()
        torch.cuda.synchronize()
        for _, off, c_ in pieces:
            assert torch.equal(xb[off:off + c_], xa[off:off + c_]), "transposed-exchange piece through RCCL"
        for off in own:
            assert bool((xb[off:off + cnt] == 0).all())
        report["stage"] = "C"

        # ---- E: the partitioned multiply over RCCL, looped back
        from oracle import oracle as orc
        from gpu_util import orc_msc, orc_sub
        L = 20
        sub = Full(L=L)
        Hm = models.mbl(L)
        Hm.establish_L()
        Hm.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(Hm.msc)
        for P, me in ((2, 0), (8, 5)):
            c = sub._c()
            c.vec_swizzle = 10                        # small blocks: a shift that really permutes them
            n_loc = (1 << L) // P
            h = backend.create_mat(masks, offs, Hm.msc['signs'], Hm.msc['coeffs'], c, c, False, 0, me, P)
            mat = backend.ShellMat(h, c, c, P, me)
            assert mat.n_local == n_loc and mat.recvs and mat.partners
            # what every partner q sends to this rank: q's own list (host-only handle of rank q), in the order of
            # this rank's receives -- all peers are this process, so posting order is matching order
            psends = {}
            for q in mat.partners:
                hq = backend.create_mat(masks, offs, Hm.msc['signs'], Hm.msc['coeffs'], c, c, False, _lib.MAT_HOST_ONLY, q, P)
                sq, _ = backend.exchange_plan(hq)
                psends[q] = [t for t in sq if t[0] == me]
                _lib.check(_lib.lib().dnm_mat_destroy(hq))
            taken = {q: 0 for q in mat.partners}
            loop_sends = []
            for q, off, cnt in mat.recvs:
                _, soff, scnt = psends[q][taken[q]]
                taken[q] += 1
                assert scnt == cnt
                loop_sends.append((0, soff, scnt))
            assert all(taken[q] == len(psends[q]) for q in mat.partners)
            mat.sends = loop_sends
            mat.recvs = [(0, off, cnt) for _, off, cnt in mat.recvs]
            x0 = rs.standard_normal(n_loc) + 1j * rs.standard_normal(n_loc)
            xv, yv = backend.Vec(n_loc, swz=10), backend.Vec(n_loc, swz=10)
            xv.set_local_from_numpy(x0)
            yv.set(3.0)
            mat.mult(xv, yv)
            torch.cuda.synchronize()
            ref = orc.matvec(orc_msc(Hm), orc_sub(sub), orc_sub(sub), np.tile(x0, P), nthreads=4)[me * n_loc:(me + 1) * n_loc]
            err = float(np.abs(yv.local_numpy() - ref).max())
            assert err < 1e-12, "partitioned multiply over RCCL (rank %d of %d): %.3e" % (me, P, err)
            mat.destroy()
        report["stage"] = "E"

        # ---- F: the same multiply as ONE native call (dnm_mat_mult_partitioned) through ShellMat.mult: the library's
        # own communicator (sharing the RCCL this process already holds), made to stand for rank 5 of 8
        import ctypes as C
        comm = backend.native_comm()
        P, me = 8, 5
        c = sub._c()
        c.vec_swizzle = 10
        n_loc = (1 << L) // P
        h = backend.create_mat(masks, offs, Hm.msc['signs'], Hm.msc['coeffs'], c, c, False, 0, me, P)
        mat = backend.ShellMat(h, c, c, P, me)
        x0 = rs.standard_normal(n_loc) + 1j * rs.standard_normal(n_loc)
        xv, yv = backend.Vec(n_loc, swz=10), backend.Vec(n_loc, swz=10)
        xv.set_local_from_numpy(x0)
        px = (C.c_void_p * P)(*[xv.array.data_ptr()] * P)           # all rank blocks equal: every peer's block is this one
        _lib.check(_lib.lib().dnm_comm_loopback(comm, me, P, px, None))
        mat._native = comm
        yv.set(3.0)
        mat.mult(xv, yv)
        torch.cuda.synchronize()
        ref = orc.matvec(orc_msc(Hm), orc_sub(sub), orc_sub(sub), np.tile(x0, P), nthreads=4)[me * n_loc:(me + 1) * n_loc]
        err = float(np.abs(yv.local_numpy() - ref).max())
        assert err < 1e-12, "native partitioned multiply (rank %d of %d): %.3e" % (me, P, err)
        mat.destroy()
        report["stage"] = "F"
    except AssertionError:
        raise
    except Exception as e:          # RCCL's own refusal (recorded in DESIGN.md section 6 if it ever shows up)
        report["refused"] = "%s: %s" % (type(e).__name__, str(e)[:600])

    # ---- D: the solvers with their hooks reducing through RCCL (world size 1: the hooks are forced on)
    L = 14
    config.L = L
    config._initialize()
    sub, H = Full(L=L), models.mbl(L)
    H.add_subspace(sub)
    x = State(subspace=sub, state='random', seed=3)
    z0 = H.evolve(x, t=0.4)
    e0 = H.eigsolve(nev=1, tol=1e-10)
    mat = H.get_mat(subspaces=(sub, sub)) if hasattr(H, "get_mat") else None
    if mat is not None:
        assert mat.exchange_only(x.vec) is None            # a world-size-1 operator has nothing to exchange
    calls = {"n": 0}
    real_allreduce = dist.all_reduce

    def counting(tensor, *a, **k):
        calls["n"] += 1
        assert tensor.is_cuda
        return real_allreduce(tensor, *a, **k)
    dist.all_reduce = counting
    backend._dist = lambda: dist
    computations._dist = lambda: dist
    config.native_comm = False          # (the HOST hooks are what this stage counts; on RCCL the native ones are the default)
    try:
        z1 = H.evolve(x, t=0.4)
        e1 = H.eigsolve(nev=1, tol=1e-10)
    finally:
        dist.all_reduce = real_allreduce
    assert calls["n"] > 0, "the hooks did not reduce through torch.distributed"
    assert np.max(np.abs(z1.to_numpy() - z0.to_numpy())) < 1e-12
    assert abs(e1[0] - e0[0]) < 1e-9
    report["hook_allreduces"] = calls["n"]
    report["stage"] += "D"
    dist.barrier()
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()
    print(json.dumps(report))
    return 77 if report["refused"] else 0


if __name__ == "__main__":
    sys.exit(main())
