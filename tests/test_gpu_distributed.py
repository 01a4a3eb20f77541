"""
End-to-end runs of the partitioned path with several ranks on ONE GPU.

RCCL refuses two ranks on one device, so these process groups use the gloo backend; dynamite_amd/_comm.py then
stages the exchanged blocks through host memory.  Everything else is the production path: every rank builds
its plan, runs the rank-local and partner passes of the HIP kernels on its block, the Krylov solvers reduce
their scalars through the hooks, SpinConserve ranks assemble their column windows.  Results are compared on
rank 0 with the oracle / scipy on the gathered vectors.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), RANK=str(rank),
                      WORLD_SIZE=str(world), DNM_TILE_BITS="8", DNM_LOG_ROWS="2", DNM_PLAN_MODE="2", DNM_GBITS="3",
                      DNM_AMIN="3")
    # several rank processes on ONE GPU: with the default of four hardware queues per process, four processes and more
    # oversubscribe the device's queue slots and every cross-stream event wait costs a scheduler time slice (a native
    # 4-rank case: 75 s against 7 s) -- an artefact of sharing the device, not of the schedules
    if not os.environ.get("DNM_TEST_NO_QUEUE_DEFAULT"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    if case == "sc":
        os.environ["DNM_SC_BLOCK"] = "10"
    # small vectors: a swizzle shift that really permutes them (conftest's choice for the one-process GPU tests)
    os.environ.setdefault("DNM_SWZ", os.environ.get("DNM_TEST_SWZ", "6"))
    if case == "full_partner":
        os.environ["DNM_EXCHANGE"] = "partner"          # four ranks would take the transposed exchange
    elif case == "full_transpose":
        os.environ["DNM_EXCHANGE"] = "transpose"        # two ranks would take the partner blocks
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("DNM_TEST_HANG_S", "600")), exit=True)   # a stuck rank reports where
    import datetime
    import torch.distributed as dist
    # (generous: on a fresh box the ranks page the image in at different speeds -- the first import of torch and of the
    # HIP libraries takes a minute or two -- and a rank that arrives late must not time its peers out)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=900))
    import scipy.sparse.linalg as spla
    from dynamite_amd import config, models
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import Full, Parity, SpinConserve
    from dynamite_amd.computations import reduced_density_matrix, entanglement_entropy
    from oracle import oracle as orc
    from gpu_util import orc_msc, orc_sub

    L = 24 if case == "sc_big" else (12 if case == "full_syk" else 14)
    config.L = L
    config._initialize()
    if case in ("sc3", "sc3_graph"):
        # SpinConserve states in the internal three-field layout (csrc/sc3.h): ranks own whole blocks of equal top
        # bits, windows and the exchange are expressed in positions of the layout (small kernel instances (6, 4))
        config.sc_layout, config.sc_layout_min_dim = (6, 4), 0
    if case == "sc_big":
        # the default SpinConserve path of BASELINE config 5: 13-bit blocks, equal-size block order, blocks cut by
        # the ownership boundaries, column windows -- multiply only (2.7 M rows)
        os.environ.pop("DNM_SC_BLOCK", None)
        sub, H = SpinConserve(L, L // 2), models.heisenberg(L)
        H.add_subspace(sub)
        x = State(subspace=sub, state='random', seed=3)
        assert "block form (13" in H.get_mat().describe()
        y = H.dot(x)
        xg, yg = x.to_numpy(to_all=True), y.to_numpy(to_all=True)
        ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xg, nthreads=2)
        assert np.max(np.abs(yg - ref)) < 1e-12, "partitioned SpinConserve multiply (default block kernel)"
        summ = H.get_mat().exchange_summary()          # only the ranges of the window that are read travel
        assert summ["scheme"] == "window" and 0 < summ["bytes_in"] < summ["window_bytes"]
        # (three ranks: the middle rank's window has no holes; config 5's 8-rank share in test_gpu_fullsize.py has)
        assert summ["bytes_in"] <= summ["window_bytes"] - 16 * H.get_mat().m_local, summ
        # the rows that read only the rank's own block ran under the exchange (dnm_mat_window_local_rows): the first
        # and the last rank own long stretches of equal top bits
        # (a host-schedule attribute; the native schedule makes the same split inside the library)
        local, remote = H.get_mat()._row_ranges or ([], [])
        if rank in (0, world - 1) and H.get_mat()._native is None:
            assert local and remote and sum(b - a for a, b in local) > 0.05 * H.get_mat().m_local, (local, remote)
        z = H.evolve(x, t=0.3, algo='chebyshev')
        assert abs(z.norm() - 1) < 1e-9 and abs(z.dot(H.dot(z)).imag) < 1e-9
        # a known answer beyond what a dense solve reaches, through the partitioned solver: 0.25 sum (XX + YY) on the same
        # subspace and partition is a chain of free fermions; its ground state the filled Fermi sea
        from dynamite_amd.operators import sigmax, sigmay, op_sum
        Hx = op_sum(0.25 * (sigmax(i) * sigmax(i + 1) + sigmay(i) * sigmay(i + 1)) for i in range(L - 1))
        Hx.L = L
        Hx.add_subspace(sub)
        e0 = Hx.eigsolve(nev=1, tol=1e-10, subspace=sub)[0]
        exact = np.sort(np.cos(np.pi * np.arange(1, L + 1) / (L + 1)))[:L // 2].sum()
        assert abs(e0 - exact) < 1e-8 * abs(exact), (e0, exact)
        _native_ran(H, case)
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        if rank == 0:
            open(os.path.join(out_dir, "ok_%s_%d" % (case, world)), "w").write("ok")
        dist.destroy_process_group()
        return
    if case.startswith("xparity"):
        # XParity on top of a product-state subspace on several ranks (the reference halves the local sizes of the parent's
        # partition, bpetsc_template_2.c:223-230, and multiplies with the operator XParity.reduce_msc wrote,
        # subspaces.py:632-674): the flip-composed hops reach the mirror image of a rank's block.  Oracle: the reduced
        # operator (the Python layer's reduce_msc is pinned by tests/golden/xparity.npz) through orc.xparity(parent).
        from dynamite_amd.subspaces import XParity
        from dynamite_amd import msc_tools
        import scipy.sparse.linalg as spla2
        parent = Full(L=L) if case == "xparity_full" else SpinConserve(L, L // 2)
        for sector in ('+', '-'):
            sub = XParity(parent, sector=sector)
            H = models.heisenberg(L) if case == "xparity_sc" else models.ising(L)
            H.add_subspace(sub)
            x = State(subspace=sub, state='random', seed=3)
            y = H.dot(x)
            xg, yg = x.to_numpy(to_all=True), y.to_numpy(to_all=True)
            assert xg.shape == (sub.get_dimension(),) and abs(np.linalg.norm(xg) - 1) < 1e-12
            H.establish_L()
            H.reduce_msc()
            m = sub.reduce_msc(H.msc)
            masks, offs = msc_tools.get_mask_offsets(m)
            osub = orc.xparity(orc_sub(parent))
            ref = orc.matvec(orc.Msc(masks, offs, m['signs'], m['coeffs']), osub, osub, xg)
            assert np.max(np.abs(yg - ref)) < 1e-12, "partitioned XParity multiply (%s, sector %s)" % (case, sector)
            assert abs(x.dot(y) - np.vdot(xg, yg)) < 1e-12
            Hs = H.to_numpy(subspaces=(sub, sub), sparse=True)
            z = H.evolve(x, t=0.4)
            want = spla2.expm_multiply(-0.4j * Hs, xg)
            assert np.max(np.abs(z.to_numpy(to_all=True) - want)) < 1e-8, "partitioned XParity evolve"
            ev = H.eigsolve(nev=2, tol=1e-10, subspace=sub)
            low = np.sort(spla2.eigsh(Hs, k=2, which='SA', tol=1e-12, return_eigenvectors=False))
            assert np.max(np.abs(np.array(ev[:2]) - low)) < 1e-8, "partitioned XParity eigsolve"
            if case == "xparity_full":
                # ... and in real arithmetic (DNM_MAT_REAL_PACKED under XParity on a power-of-two number of ranks)
                from dynamite_amd.computations import eigsolve as _eig
                os.environ["DNM_EIGS_REAL"] = "1"
                er, vr = H.eigsolve(nev=2, tol=1e-10, subspace=sub, getvecs=True)
                os.environ.pop("DNM_EIGS_REAL")
                assert _eig.last_stats['real_arithmetic'] is True
                assert np.max(np.abs(np.array(er[:2]) - low)) < 1e-8, "partitioned XParity eigsolve, real arithmetic"
                vg = vr[0].to_numpy(to_all=True)
                assert np.abs(vg.imag).max() == 0.0 and np.linalg.norm(Hs @ vg - er[0] * vg) < 1e-7
            _native_ran(H, case)
            H.destroy_mat()
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        if rank == 0:
            open(os.path.join(out_dir, "ok_%s_%d" % (case, world)), "w").write("ok")
        dist.destroy_process_group()
        return
    if case in ("explicit", "auto", "projection", "full_odd", "parity_odd"):
        # partitions that are not XOR-partner exchanges: rows in index order, columns through a window
        # (the reference runs these through MatMult_CPU_General's MPI branch, bpetsc_template_2.c:413-504,
        # exercised at 3 ranks by tests/integration/run_all_tests.py:109-113)
        from dynamite_amd.subspaces import Explicit, Auto
        H = models.mbl(L) if case != "auto" else models.heisenberg(L)
        if case == "explicit":
            rs = np.random.RandomState(5)
            left = right = Explicit(np.sort(rs.choice(1 << L, size=3001, replace=False)), L=L)
        elif case == "auto":
            left = right = Auto(H, 'U' * (L // 2) + 'D' * (L - L // 2))
        elif case == "projection":
            left, right = Parity('even', L=L), Full(L=L)
        elif case == "full_odd":
            left = right = Full(L=L)
        else:
            left = right = Parity('odd', L=L)
        H.allow_projection = True
        H.add_subspace(left, right)
        # small blocks: let the row-range split of the window multiply run whatever the length of the ranges
        from dynamite_amd import backend as _be
        _be.ShellMat.WINDOW_ROWS_MIN_BLOCKS, _be.ShellMat.WINDOW_ROWS_MIN_SHARE = 1, 0.0
        x = State(subspace=right, state='random', seed=3)
        y = H.dot(x, result=State(subspace=left))
        assert y.subspace == left and "tiled=1" not in H.get_mat(subspaces=(left, right)).describe()
        xg, yg = x.to_numpy(to_all=True), y.to_numpy(to_all=True)
        ref = orc.matvec(orc_msc(H), orc_sub(left), orc_sub(right), xg)
        assert np.max(np.abs(yg - ref)) < 1e-12, "partitioned multiply through a column window (%s)" % case
        if left is right:
            import scipy.sparse.linalg as spla2
            Hs = H.to_numpy(subspaces=(left, right), sparse=True)
            z = H.evolve(x, t=0.4)
            want = spla2.expm_multiply(-0.4j * Hs, xg)
            assert np.max(np.abs(z.to_numpy(to_all=True) - want)) < 1e-8, "evolve on a window partition"
            ev = H.eigsolve(nev=1, tol=1e-10, subspace=left)
            low = spla2.eigsh(Hs, k=1, which='SA', tol=1e-12, return_eigenvectors=False)
            assert abs(ev[0] - low[0]) < 1e-8, "eigsolve on a window partition"
        _native_ran(H, case)
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        if rank == 0:
            open(os.path.join(out_dir, "ok_%s_%d" % (case, world)), "w").write("ok")
        dist.destroy_process_group()
        return
    if case in ("full", "full_partner", "full_transpose"):
        sub, H = Full(L=L), models.mbl(L)
    elif case == "full_syk":
        # masks of many terms as table records (csrc/plan.h DevTab) on several ranks: flipped bits among the rank bits, the
        # partners' blocks as the source of the gathers
        sub, H = Full(L=L), models.syk(L)
    elif case == "parity":
        sub, H = Parity('even', L=L), models.mbl(L)
    elif case == "sc3_graph":
        # an operator on a bond graph (a J1-J2 ring with two long bonds) in the internal layout on several ranks: the
        # bond-graph passes of csrc/sc3g_kernels.hip read their gathered hops through the column window (no site
        # relabelling on partitions: a rank owns whole blocks of equal top bits of the reference's labelling)
        edges = [(i, (i + 1) % L) for i in range(L)] + [(i, (i + 2) % L) for i in range(0, L, 2)] + [(0, 7), (3, 11)]
        sub, H = SpinConserve(L, L // 2), models.bond_heisenberg([(min(a, b), max(a, b)) for a, b in edges], L=L)
    else:
        sub, H = SpinConserve(L, L // 2), models.mbl(L)
    H.add_subspace(sub)
    dim = sub.get_dimension()

    x = State(subspace=sub, state='random', seed=3)
    start, end = x.vec.getOwnershipRange()
    from dynamite_amd.backend import split_ownership
    if case in ("sc3", "sc3_graph"):
        assert x.vec.internal and "internal layout" in H.get_mat().describe()
        assert ("bond graph" in H.get_mat().describe()) == (case == "sc3_graph")
        import torch
        t = torch.tensor([start, end], dtype=torch.int64)
        allr = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allr, t)
        assert int(allr[0][0]) == 0 and int(allr[-1][1]) == dim
        assert all(int(allr[q][1]) == int(allr[q + 1][0]) for q in range(world - 1))     # contiguous in reference order
    else:
        assert (start, end - start) == split_ownership(dim, world, rank)
    xg = x.to_numpy(to_all=True)
    assert xg.shape == (dim,) and abs(np.linalg.norm(xg) - 1) < 1e-12
    assert abs(x.norm() - 1) < 1e-12

    # multiply
    if case not in ("sc", "sc3", "sc3_graph"):
        # exchange scheme: partner blocks on two ranks, the transposed all-to-all from four on (backend.py)
        want_scheme = {"full": "transpose" if world >= 4 else "partner", "full_partner": "partner",
                       "full_syk": "partner",      # (all-to-all masks: nothing for the transposed exchange to gain)
                       "full_transpose": "transpose", "parity": "transpose" if world >= 4 else "partner"}[case]
        assert H.get_mat().exchange_summary()["scheme"] == want_scheme
        if case == "full_syk":
            assert "table records: " in H.get_mat().describe() or want_scheme == "transpose", H.get_mat().describe()
        if want_scheme == "transpose" and case in ("full", "parity"):
            # ... and runs sub-piece by sub-piece (forward parts, ranges of the layout-B pass, returns)
            assert H.get_mat()._tr_pipe or H.get_mat()._native_tr, "the transposed exchange should pipeline at this size"
    y = H.dot(x)
    yg = y.to_numpy(to_all=True)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xg, nthreads=2)
    assert np.max(np.abs(yg - ref)) < 1e-12, "partitioned multiply"
    assert abs(x.dot(y) - np.vdot(xg, yg)) < 1e-12           # <x|y> as Vec.dot (conjugates x)
    assert abs(H.infinity_norm(subspaces=(sub, sub)) - orc.infnorm(orc_msc(H), orc_sub(sub), orc_sub(sub))) < 1e-12
    # gather to rank 0 only
    y0 = y.to_numpy()
    assert (y0 is None) == (rank != 0)

    if case == "full_syk":
        # (the multiply is what the table records change; SYK's all-to-all masks make every solver step an exchange with
        # every other rank, minutes of host staging on one shared GPU: the solvers run on the other cases)
        z = H.evolve(x, t=0.05)
        assert abs(z.norm() - 1) < 1e-9
        _native_ran(H, case)
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        if rank == 0:
            open(os.path.join(out_dir, "ok_%s_%d" % (case, world)), "w").write("ok")
        dist.destroy_process_group()
        return
    # Krylov solvers through the hooks
    Hs = H.to_numpy(subspaces=(sub, sub), sparse=True)
    z = H.evolve(x, t=0.6)
    zg = z.to_numpy(to_all=True)
    want = spla.expm_multiply(-0.6j * Hs, xg)
    assert np.max(np.abs(zg - want)) < 1e-8, "partitioned evolve"
    # (eight ranks on one GPU: the multiply, the norm, one evolve and one eigsolve -- the schedules that differ with
    # the rank count; the other solver variants run at two and four ranks)
    light = world >= 8 or (world >= 4 and os.environ.get("DNM_NATIVE_COMM") == "1")
    if not light:
        zc = H.evolve(x, t=0.6, algo='chebyshev').to_numpy(to_all=True)
        assert np.max(np.abs(zc - want)) < 1e-8, "partitioned Chebyshev evolve"
    evals, evecs = H.eigsolve(nev=2, getvecs=True, tol=1e-10, subspace=sub)
    lowest = np.sort(spla.eigsh(Hs, k=2, which='SA', tol=1e-12, return_eigenvectors=False))
    assert np.max(np.abs(np.array(evals[:2]) - lowest)) < 1e-8, "partitioned eigsolve"
    v0 = evecs[0].to_numpy(to_all=True)
    assert np.linalg.norm(Hs @ v0 - evals[0] * v0) < 1e-7
    if light:
        _native_ran(H, case)
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        if rank == 0:
            open(os.path.join(out_dir, "ok_%s_%d" % (case, world)), "w").write("ok")
        dist.destroy_process_group()
        return
    os.environ["DNM_EIGS_BASISFREE"] = "1"        # without a stored basis, through the hooks: the second pair by deflation
    e1, v1 = H.eigsolve(nev=2, getvecs=True, tol=1e-10, subspace=sub)
    os.environ.pop("DNM_EIGS_BASISFREE")
    assert np.max(np.abs(np.array(e1[:2]) - lowest)) < 1e-8
    for e_, v_ in zip(e1[:2], v1[:2]):
        v1g = v_.to_numpy(to_all=True)
        assert np.linalg.norm(Hs @ v1g - e_ * v1g) < 1e-7

    if case in ("full", "full_partner", "full_transpose", "parity"):
        # real arithmetic on a partitioned Full / Parity operator: the packed operator (bit 0 of the index = the lane)
        # exchanges like an operator on one bit less -- partner blocks on two ranks, the transposed exchange from four on
        # (or as DNM_EXCHANGE says), half the bytes either way
        from dynamite_amd.computations import eigsolve as _eig
        os.environ["DNM_EIGS_REAL"] = "1"
        er, vr = H.eigsolve(nev=2, getvecs=True, tol=1e-10, subspace=sub)
        os.environ.pop("DNM_EIGS_REAL")
        assert _eig.last_stats['real_arithmetic'] is True
        pm = H.get_real_packed_mat(sub)
        assert pm.real_packed and pm.exchange_summary()["scheme"] == H.get_mat().exchange_summary()["scheme"]
        assert np.max(np.abs(np.array(er[:2]) - lowest)) < 1e-8, "partitioned eigsolve, real arithmetic"
        vg = vr[0].to_numpy(to_all=True)
        assert np.abs(vg.imag).max() == 0.0 and np.linalg.norm(Hs @ vg - er[0] * vg) < 1e-7
    if case in ("sc3", "sc3_graph"):
        # the same solves in real arithmetic (DNM_MAT_REAL_PACKED on the partitioned internal layout: one double per
        # position, windows and exchange in pairs of positions -- half the bytes on the links)
        from dynamite_amd.computations import eigsolve as _eig
        os.environ["DNM_EIGS_REAL"] = "1"
        er, vr = H.eigsolve(nev=2, getvecs=True, tol=1e-10, subspace=sub)
        assert _eig.last_stats['real_arithmetic'] is True
        assert np.max(np.abs(np.array(er[:2]) - lowest)) < 1e-8, "partitioned eigsolve, real arithmetic"
        vg = vr[0].to_numpy(to_all=True)
        assert np.abs(vg.imag).max() == 0.0 and np.linalg.norm(Hs @ vg - er[0] * vg) < 1e-7
        os.environ["DNM_EIGS_BASISFREE"] = "1"
        e2, v2 = H.eigsolve(nev=1, getvecs=True, tol=1e-10, subspace=sub)
        os.environ.pop("DNM_EIGS_BASISFREE")
        os.environ.pop("DNM_EIGS_REAL")
        assert _eig.last_stats['real_arithmetic'] is True and abs(e2[0] - lowest[0]) < 1e-8
        v2g = v2[0].to_numpy(to_all=True)
        assert np.linalg.norm(Hs @ v2g - e2[0] * v2g) < 1e-7

    if case == "sc3":
        # The partition made for the exchange (dnm_subspace.vec_swizzle bits 16-19 = 1, csrc/sc3.h): the T blocks in an order
        # whose contiguous ranges cut one bond of the chain.  A rank's share is no range of the reference order, so the
        # global vector goes through the WHOLE-vector map on every rank (test size) and is cut by positions of the layout.
        import ctypes as _C
        import torch
        from dynamite_amd import backend as _b, _lib as _l, _comm as _cm
        from gpu_util import marshal
        desc = sub._to_c()
        d1 = type(desc['data']).from_buffer_copy(desc['data'])
        d1.vec_swizzle = int(d1.vec_swizzle) | (1 << 16)
        sd = {'type': desc['type'], 'data': d1, '_keep': desc}
        m1 = _b.build_mat(*marshal(H), sd, sd, site_perm=False)
        assert "internal layout" in m1.describe() and m1.swz_right >> 16 == 1
        nint = _C.c_int64()
        _l.check(_l.lib().dnm_vec_layout_size(_C.byref(d1), _C.byref(nint)))
        whole = _l.Partition(0, 1)
        full = torch.zeros(nint.value, dtype=torch.complex128, device=config.device)
        nat = torch.from_numpy(xg).to(config.device)
        _l.check(_l.lib().dnm_vec_layout_copy(_C.byref(d1), _C.byref(whole), _C.c_void_p(full.data_ptr()),
                                              _C.c_void_p(nat.data_ptr()), 1, None))
        shares = [_b.layout_partition(d1, world, q) for q in range(world)]
        i0, il = shares[rank][0], shares[rank][1]
        assert sum(s_[1] for s_ in shares) == nint.value and all(s_[1] > 0 for s_ in shares)
        assert [s_[2] for s_ in shares[1:]] == [-1] * (world - 1)          # no reference side beyond the first share
        x1 = _b.Vec(dim, array=full[i0:i0 + il].clone(), swz=m1.swz_right, sub_c=d1)
        y1 = _b.Vec(dim, swz=m1.swz_left, sub_c=d1)
        m1.mult(x1, y1)
        if os.environ.get("DNM_NATIVE_COMM") == "1":
            assert m1._native is not None
        parts = _cm.gather_varied(y1.array, [s_[1] for s_ in shares], dst=0)
        if rank == 0:
            yfull = torch.cat(parts)
            ynat = torch.empty(dim, dtype=torch.complex128, device=config.device)
            _l.check(_l.lib().dnm_vec_layout_copy(_C.byref(d1), _C.byref(whole), _C.c_void_p(ynat.data_ptr()),
                                                  _C.c_void_p(yfull.data_ptr()), 0, None))
            assert np.max(np.abs(ynat.cpu().numpy() - ref)) < 1e-12, "multiply on the partition made for the exchange"
        # ... and it receives no more than the reference-compatible partition does
        s1, s0 = m1.exchange_summary(), H.get_mat().exchange_summary()
        assert s1["scheme"] == "window" and s1["bytes_in"] <= s0["bytes_in"] * 1.5 + 16 * 64
        m1.destroy()
        # eigenvalues alone are solved there (Operator.get_solver_mat) whenever its blocks can be shared out
        from dynamite_amd.computations import eigsolve as _eig
        for real in ("0", "1"):
            os.environ["DNM_EIGS_REAL"] = real
            e_only = H.eigsolve(nev=2, tol=1e-10, subspace=sub)
            os.environ.pop("DNM_EIGS_REAL")
            assert np.max(np.abs(np.array(e_only[:2]) - lowest)) < 1e-8, "eigenvalues on the solver's own partition"
            sm = H.get_solver_mat(sub, real == "1")
            assert (_eig.last_mat is sm) == (sm is not None)
            if sm is not None:
                assert sm.swz_right >> 16 == 1 and sm.real_packed == (real == "1")
        if rank == 0:
            print("solver partition in use: %s" % (H.get_solver_mat(sub, False) is not None), flush=True)
        # ... and so are the solves that hand vectors back: the state moves there block by block, the result moves back
        # (backend.reorder_blocks) -- same numbers as on the reference-compatible partition
        from dynamite_amd.computations import evolve as _ev
        sm = H.get_solver_mat(sub, False)
        if sm is not None:
            za = H.evolve(x, t=0.6)
            assert _ev.last_mat is sm
            zc2 = H.evolve(x, t=0.6, algo='chebyshev')
            assert _ev.last_mat is sm
            config.sc_solver_partition = False
            zb = H.evolve(x, t=0.6)
            assert _ev.last_mat is H.get_mat()
            config.sc_solver_partition = True
            za, zb, zc2 = (v_.to_numpy(to_all=True) for v_ in (za, zb, zc2))
            # (two partitions sum the Krylov dot products in different orders: the same answer within the solver's tolerance)
            errs = (np.max(np.abs(za - zb)), np.max(np.abs(zc2 - want)), np.max(np.abs(za - want)))
            assert errs[0] < 1e-9 and errs[1] < 1e-8 and errs[2] < 1e-8, "evolve on the solver's partition: %r" % (errs,)
            for real in ("0", "1"):
                os.environ["DNM_EIGS_REAL"] = real
                ev_, vv_ = H.eigsolve(nev=2, getvecs=True, tol=1e-10, subspace=sub)
                os.environ.pop("DNM_EIGS_REAL")
                assert _eig.last_mat is H.get_solver_mat(sub, real == "1")
                assert np.max(np.abs(np.array(ev_[:2]) - lowest)) < 1e-8
                for e_, v_ in zip(ev_[:2], vv_[:2]):
                    vg_ = v_.to_numpy(to_all=True)
                    assert np.linalg.norm(Hs @ vg_ - e_ * vg_) < 1e-7, "eigenvectors moved back from the solver's partition"

    # reduced density matrix / entropy of the partitioned state
    for keep in ([0, 1, 2], [L - 3, L - 2], [1, 5, L - 1]):
        rho = reduced_density_matrix(z, keep)
        if rank == 0:
            want_rho = orc.rdm(orc_sub(sub), zg, np.array(keep, dtype=np.int64))
            assert np.max(np.abs(rho - want_rho)) < 1e-12, "partitioned RDM"
        else:
            assert rho.shape == (1, 1)
    s = entanglement_entropy(z, list(range(L // 2)))
    if rank == 0:
        assert s > 0

    # files written by all ranks, read back
    fn = os.path.join(out_dir, "state_" + case)
    z.save(fn)
    z2 = State.from_file(fn)
    assert np.array_equal(z2.to_numpy(to_all=True), zg)

    _native_ran(H, case)
    dist.barrier()
    faulthandler.cancel_dump_traceback_later()
    if rank == 0:
        open(os.path.join(out_dir, "ok_%s_%d" % (case, world)), "w").write("ok")
    dist.destroy_process_group()


def _native_ran(H, case):
    """Under DNM_NATIVE_COMM=1 (test_native_schedule_between_rank_processes) every multiply and solver step of the
    cases the native schedule covers must have gone through dnm_mat_mult_partitioned -- a silent fall-back to the host
    schedules would make that test a copy of the one above."""
    if os.environ.get("DNM_NATIVE_COMM") != "1":
        return
    # (None: a form that does not apply, remembered; the real-packed handle on the reference-compatible partition is only
    # the eigenvectors' way back when the solves run on the solver's own partition: it never multiplies)
    idle = {k for k in H._mats if isinstance(k, tuple) and k[0] == 'real_packed'
            and H._mats.get(('solver', k[1], True)) is not None}
    mats = [m for k, m in H._mats.items() if m is not None and k not in idle]
    assert mats
    for m in mats:
        assert m._native is not None, "the native schedule did not run (%s)" % case
        assert m._tr is None, "the host's transposed schedule was built beside the native one (%s)" % case
    # the operators go first, then the communicator (the stand-in transport removes its mailboxes with the last rank)
    from dynamite_amd import backend
    H.destroy_mat()
    backend.release_native_comm()


FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def build_fake_rccl():
    """tests/fake_rccl/fake_rccl.cpp -> libfake_rccl.so (hipcc; also built by __graft_entry__.build())."""
    import subprocess
    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
    if not os.path.exists(FAKE_RCCL) or os.path.getmtime(FAKE_RCCL) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", FAKE_RCCL, src])
    return FAKE_RCCL


@pytest.mark.parametrize("case,world", [("full", 2), ("full", 4), ("full", 8), ("full_partner", 4), ("full_transpose", 2),
                                        ("full_syk", 2), ("full_syk", 4),
                                        ("parity", 2), ("parity", 4), ("sc", 2), ("sc", 3), ("sc3", 2), ("sc3", 3), ("sc3_graph", 2),
                                        ("sc_big", 3), ("explicit", 3), ("auto", 2), ("projection", 3),
                                        ("projection", 2), ("full_odd", 3), ("parity_odd", 3),
                                        ("xparity_full", 2), ("xparity_sc", 2), ("xparity_sc", 3)] +
                         ([("xparity_full", 4)] if os.environ.get("DNM_TEST_LARGEST") == "1" else []))
def test_partitioned_end_to_end_one_gpu(tmp_path, case, world):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok_%s_%d" % (case, world)))


@pytest.mark.parametrize("case,world", [("full", 2), ("full", 4), ("full", 8), ("parity", 4), ("sc", 3), ("sc3", 2), ("sc3", 3),
                                        ("full_syk", 2), ("sc3_graph", 2), ("sc_big", 3), ("explicit", 3), ("xparity_full", 2),
                                        ("xparity_sc", 3), ("full_odd", 3), ("projection", 2)] +
                         ([("full_partner", 4), ("full_transpose", 2), ("auto", 2), ("parity_odd", 3), ("projection", 3)]
                          if os.environ.get("DNM_TEST_LARGEST") == "1" else []))
def test_native_schedule_between_rank_processes(tmp_path, monkeypatch, case, world):
    """The NATIVE schedule (dnm_mat_mult_partitioned, dnm_comm_hooks -- the default on RCCL transports) between real rank
    processes.  RCCL refuses two ranks on one device, so on this one-GPU box the library binds a stand-in for librccl
    (tests/fake_rccl: mailboxes in /dev/shm, host-staged copies, message sizes CHECKED) through DNM_RCCL_LIB: every rank is
    its own process with its own handle and message lists -- partner blocks, the pipelined transposed exchange, column
    windows with their all-gather of needs, the all-reduces of the solver hooks -- and the checks are those of
    test_partitioned_end_to_end_one_gpu: multiply against the oracle, evolve / eigsolve against scipy, real arithmetic,
    reduced density matrices.  What this cannot show (asynchronous RCCL kernels against the compute stream) is what the
    loop-back tests with the real RCCL show (test_native_partitioned_multiply).  Replaces bpetsc_template_2.c:413-504,
    787-879 / bcuda_template_2.cu:161-171."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("DNM_RCCL_LIB", build_fake_rccl())
    monkeypatch.setenv("DNM_NATIVE_COMM", "1")
    monkeypatch.setenv("DNM_FAKE_RCCL_TIMEOUT_S", "300")
    before = {d for d in os.listdir("/dev/shm") if d.startswith("dnmfake_")}      # (what an earlier, failed run left)
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok_%s_%d" % (case, world)))
    left = {d for d in os.listdir("/dev/shm") if d.startswith("dnmfake_")} - before
    assert not left, "mailboxes left behind: %r" % left


def _bench_ranks(world, extra_env, argv=(), L=22, timeout=900):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one rank per 'GPU'), with the ranks
    sharing this GPU over gloo (DNM_BENCH_BACKEND)"""
    import json
    import subprocess
    env = dict(os.environ, DNM_BENCH_BACKEND="gloo", **extra_env)
    env.setdefault("GPU_MAX_HW_QUEUES", "2")        # ranks share one GPU (see _worker)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--L", str(L), "--config5", "18,9"] + list(argv)
    # a session of its own: a run that exceeds the limit is ended as a whole (the launcher AND its ranks -- orphaned ranks
    # would keep their device memory for the rest of the suite)
    import signal
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        stdout, stderr = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        stdout, stderr = p.communicate()
        raise AssertionError("bench.py did not finish within %d s\n%s" % (timeout, stderr[-3000:]))
    assert p.returncode == 0, stderr[-3000:]
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0]), stderr


def _check_schedule_entry(e, world):
    assert e["ms_per_step"] > 0 and e["exchange_only_ms"] > 0 and e["compute_only_ms"] > 0
    assert e["busiest_link_bytes"] > 0 and e["link_GBs_measured"] > 0
    assert abs(e["hidden_ms"] - (e["exchange_only_ms"] + e["compute_only_ms"] - e["ms_per_step"])) < 1e-9
    assert e["selfcheck"].startswith("sampled rows"), e


def _check_config5(sec):
    c5 = sec["config5"]
    assert c5["dim"] == 48620 and "SpinConserve(18,9)" in c5["workload"]
    h, kx = c5["heisenberg"], c5["known_answer_xx_chain"]
    assert h["exchange"] == "window" and h["matvecs"] > 10 and h["measured_rel_residual"] <= 1.01e-8
    assert h["bytes_received_per_multiply_rank0"] > 0 and "failed_checks" not in h and "partition" in h
    m = h["multiply"]
    assert m["ms"] > 0 and m["exchange_only_ms"] > 0 and m["compute_only_ms"] > 0, m
    assert kx["abs_error"] < 1e-6 and "failed_checks" not in kx


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_flow_one_gpu(world):
    """The N > 1 line on gloo-staged ranks (host schedule; RCCL refuses two ranks on one device): the exchange plan,
    barriers, max-over-ranks timing, the multiply split into exchange / compute / hidden, the check of the first multiply
    against the MSC definition, config 5's eigsolve at a toy size -- a plumbing check of the multi-rank flow, not a
    measurement."""
    d, _ = _bench_ranks(world, {})
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "Gamplitudes/s"
    assert d["value"] > 0 and d["scaling"] == "weak" and d["config"]["L"] == 22
    assert d["config"]["launches_per_step"] >= 3          # rank-local passes plus partner / transposed-layout passes
    assert d["config"]["exchange"] == ("transpose" if world >= 4 else "partner")
    assert d["config"]["xgmi_busiest_link_bytes"] > 0
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm"
    # the link rate measured in the run sits next to the assumed one; the first multiply was checked
    assert d["config"]["xgmi_link_GBs_measured"] > 0 and d["config"]["xgmi_exchange_only_ms"] > 0
    assert d["config"]["exchange_selfcheck"].startswith("sampled rows")
    assert d["config"]["schedule"].startswith("host")
    mg = d["multi_gpu"]
    assert mg["default_schedule"] == "host" and mg["first_contact_probe"] is None
    _check_schedule_entry(mg["schedules"]["host"], world)
    assert mg["schedules"]["native"].startswith("not run")
    _check_config5(d["secondary"])
    assert d["secondary_ok"] is True


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_flow_native_schedule(world):
    """The same launch with the native schedule as the default (what an RCCL transport gives; here the stand-in
    transport of tests/fake_rccl under DNM_NATIVE_COMM=1): the line times BOTH schedules, each split three ways, and
    config 5's eigsolve runs through the native hooks."""
    d, _ = _bench_ranks(world, {"DNM_NATIVE_COMM": "1", "DNM_RCCL_LIB": build_fake_rccl()})
    assert d["value"] > 0 and d["config"]["schedule"].startswith("native")
    assert d["config"]["exchange"] == ("transpose" if world >= 4 else "partner")
    mg = d["multi_gpu"]
    assert mg["default_schedule"] == "native" and mg["first_contact_probe"] is None      # (a decision already taken: no probe)
    _check_schedule_entry(mg["schedules"]["native"], world)
    _check_schedule_entry(mg["schedules"]["host"], world)
    assert mg["schedules"]["native"]["busiest_link_bytes"] == mg["schedules"]["host"]["busiest_link_bytes"]
    _check_config5(d["secondary"])
    assert d["secondary_ok"] is True


@pytest.mark.parametrize("world,native", [(4, True), (3, False)])
def test_bench_config5_on_the_partition_made_for_the_exchange(world, native):
    """config 5 of the line with its subspace in the internal layout (at BASELINE's size it is; here the toy size takes it
    through DNM_SC_LAYOUT / DNM_SC_LAYOUT_MIN_DIM): the eigsolve iterates on the partition made for the exchange and the
    line says so, with the bytes of both partitions."""
    env = {"DNM_SC_LAYOUT": "6,4", "DNM_SC_LAYOUT_MIN_DIM": "0", "DNM_EXPERIMENTAL": "1"}
    if native:
        env.update(DNM_NATIVE_COMM="1", DNM_RCCL_LIB=build_fake_rccl())
    d, _ = _bench_ranks(world, env, L=20 if world == 3 else 22)
    _check_config5(d["secondary"])
    for key in ("heisenberg", "known_answer_xx_chain"):
        h = d["secondary"]["config5"][key]
        assert h["partition"].startswith("made for the exchange (block order 1"), h["partition"]
        assert h["bytes_received_per_multiply_rank0_reference_compatible_partition"] > 0
        assert h["bytes_received_per_multiply_busiest_rank"] >= h["bytes_received_per_multiply_rank0"] > 0
        assert h["busiest_link_bytes_any_rank"] <= h["bytes_received_per_multiply_busiest_rank"]
        assert h["busiest_link_bytes_any_rank_reference_compatible_partition"] > 0
        assert h["multiply"]["schedule"] == ("native" if native else "host")
    assert d["secondary_ok"] is True


def test_bench_first_contact_probe():
    """Before a rank touches its GPU a child process goes through the native schedule at a small size (here forced onto
    the stand-in transport): passed -> the native schedule is the default of the run and the probe's report is in the
    line."""
    d, _ = _bench_ranks(2, {"DNM_BENCH_FORCE_PROBE": "1", "DNM_RCCL_LIB": build_fake_rccl()}, ["--probe-timeout", "400"])
    pr = d["multi_gpu"]["first_contact_probe"]
    assert pr["ok"] is True and pr["all_ranks_ok"] is True and pr["native"] is True, pr
    assert pr["multiply_selfcheck"]["scheme"] == "partner" and pr["eigsolve_sc26_13"]["matvecs"] > 10
    assert d["multi_gpu"]["default_schedule"] == "native" and d["config"]["schedule"].startswith("native")
    _check_schedule_entry(d["multi_gpu"]["schedules"]["host"], 2)


@pytest.mark.parametrize("how", ["fails", "hangs"])
def test_bench_first_contact_probe_saves_the_run(how):
    """A native schedule that does not come up (the stand-in transport refuses to initialise) or never returns (its first
    exchange hangs): that costs the probe's child -- ended after --probe-timeout -- and the run measures the host
    schedule and says why."""
    env = {"DNM_BENCH_FORCE_PROBE": "1", "DNM_RCCL_LIB": build_fake_rccl(),
           "DNM_FAKE_RCCL_FAIL" if how == "fails" else "DNM_FAKE_RCCL_HANG": "1"}
    d, err = _bench_ranks(2, env, ["--probe-timeout", "240" if how == "fails" else "60"])
    pr = d["multi_gpu"]["first_contact_probe"]
    assert pr["ok"] is False and pr["all_ranks_ok"] is False and pr["error"], pr
    # (a hang is ended by the child's own watchdog -- exit code 3 -- or, failing that, by the parent after --probe-timeout)
    assert ("no answer within" in pr["error"] or "exit code 3" in pr["error"]) == (how == "hangs")
    assert d["value"] > 0 and d["multi_gpu"]["default_schedule"] == "host"
    assert d["multi_gpu"]["schedules"]["native"] == "not run: the first-contact probe failed"
    _check_schedule_entry(d["multi_gpu"]["schedules"]["host"], 2)
    assert "first-contact probe of the native schedule failed" in err


def test_bench_falls_back_when_the_selfcheck_fails():
    """A transposed-exchange operator whose first multiply fails its sampled-row check (forced here): every rank gets
    the same verdict, the bench rebuilds the operator with partner blocks, times that, and says so in its line."""
    d, _ = _bench_ranks(4, {"DNM_TEST_FAIL_SELFCHECK": "1", "DNM_EXPERIMENTAL": "1"})
    assert d["config"]["exchange"] == "partner" and d["config"]["exchange_selfcheck"].startswith("failed")
    assert d["value"] > 0 and d["multi_gpu"]["schedules"]["host"]["selfcheck"].startswith("failed")


def _rccl_worker(rank, world, port, out_dir):
    """One rank per GPU over RCCL: the device-to-device branch of _comm.batch_p2p (no host staging)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), RANK=str(rank),
                      WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("DNM_TEST_HANG_S", "600")), exit=True)
    import datetime
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank),
                            timeout=datetime.timedelta(seconds=900))
    from dynamite_amd import config, models, _comm
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import Full
    from oracle import oracle as orc
    from gpu_util import orc_msc, orc_sub
    L = 20
    config.L = L
    config._initialize()
    # the transport on its own: every rank sends a tagged block to every other rank, device to device
    mine = torch.full((1 << 12,), complex(rank + 1, -rank), dtype=torch.complex128, device=config.device)
    bufs = {q: torch.empty(1 << 12, dtype=torch.complex128, device=config.device) for q in range(world) if q != rank}
    assert not _comm._staged(mine)
    for r in _comm.batch_p2p([(mine, q) for q in bufs], [(b, q) for q, b in bufs.items()]):
        r.wait()
    torch.cuda.synchronize()
    for q, b in bufs.items():
        assert bool((b == complex(q + 1, -q)).all()), "RCCL p2p block from rank %d" % q
    # and the partitioned multiply through it (partner blocks over xGMI, overlapped with the local passes)
    sub, H = Full(L=L), models.mbl(L)
    H.add_subspace(sub)
    x = State(subspace=sub, state='random', seed=3)
    y = H.dot(x)
    xg, yg = x.to_numpy(to_all=True), y.to_numpy(to_all=True)
    ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xg, nthreads=2)
    assert np.max(np.abs(yg - ref)) < 1e-12, "partitioned multiply over RCCL"
    z = H.evolve(x, t=0.3)
    assert abs(z.norm() - 1) < 1e-9
    dist.barrier()
    faulthandler.cancel_dump_traceback_later()
    if rank == 0:
        open(os.path.join(out_dir, "ok_rccl_%d" % world), "w").write("ok")
    dist.destroy_process_group()


def test_rccl_transport_when_several_gpus(tmp_path):
    """The production transport (one rank per GPU, RCCL send/recv of device memory -- what replaces the all-gather of
    bcuda_template_2.cu:161-171): runs wherever at least two GPUs are visible, so that the first multi-GPU
    benchmark is not also its first execution.  Skipped on one-GPU boxes."""
    import torch
    import torch.multiprocessing as mp
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs at least two GPUs (this box has %d)" % n)
    world = 2 if n < 4 else 4
    mp.spawn(_rccl_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok_rccl_%d" % world))


def test_rccl_transport_world_one():
    """First execution of the unstaged (device-to-device, RCCL) branch of dynamite_amd/_comm.py on a one-GPU box: a
    process group of world size 1 over the "nccl" backend whose only peer is the rank itself -- isend + irecv of
    complex128 device slices in one batch, the message lists of real partner / transposed plans, all_gather, reduce,
    the solver hooks' all-reduces, and the partitioned multiply itself (rank 0 of 2, rank 5 of 8) with its exchange
    looped back to the rank (tests/rccl_self_child.py, started as a fresh process).  If RCCL refuses a send
    to the sending rank, the collectives and the hooks must still have run and the test is skipped with RCCL's
    message.  Replaces bcuda_template_2.cu:161-171."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_self_child.py")], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode in (0, 77) and lines, "rc %d\n%s\n%s" % (out.returncode, out.stdout[-1500:], out.stderr[-3000:])
    rep = json.loads(lines[-1])
    assert rep["backend"] == "nccl" and rep["stage"].endswith("D") and rep["hook_allreduces"] > 0, rep
    if out.returncode == 77:
        assert rep["stage"] in ("AD", "BD", "CD", "ED"), rep
        pytest.skip("RCCL refused a send to the sending rank: " + str(rep["refused"]))
    assert rep["stage"] == "FD" and rep["refused"] is None, rep


def test_native_partitioned_multiply(tmp_path):
    """The partitioned multiply as one native call (dnm_comm_* / dnm_mat_mult_partitioned, csrc/comm.cpp) from a
    process that binds the C ABI with ctypes alone (no torch, no backend.py: tests/native_comm_child.py): RCCL
    communicator of one rank standing, in turn, for every rank of P (dnm_comm_loopback) -- XOR-partner blocks of a
    swizzled Full-space operator on 2 and 8 ranks, column windows of SpinConserve in reference order and of the Full
    space on 3 ranks, the two tiled SpinConserve passes split around the exchange on 2 and 3 ranks, the solver hooks --
    every rank's rows against the oracle (bpetsc_template_2.c:413-504, 787-879 are what this replaces)."""
    import json
    import subprocess
    from dynamite_amd import models
    from gpu_util import marshal, orc_msc, orc_sub, rand_state
    from dynamite_amd.subspaces import Full, SpinConserve, Parity
    from dynamite_amd import _lib
    from oracle import oracle as orc
    cases = {}

    def add(name, H, sub, P, typ, swz, exchange=0, flags=0, env=None):
        masks, offs, signs, coeffs = marshal(H)
        x = rand_state(sub.get_dimension(), seed=len(cases) + 1)
        if flags & _lib.MAT_REAL_PACKED:
            x = x.real + 0j
        y = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
        L = sub.L
        k = getattr(sub, "k", 0)
        nck = sub._nchoosek if typ == 3 else np.zeros((1, L + 1), dtype=np.int64)
        for key, val in (("masks", masks), ("mask_offsets", offs), ("signs", signs), ("coeffs", coeffs), ("x", x),
                         ("y", y), ("type", typ), ("L", L), ("k", k), ("P", P), ("swz", swz), ("nck", nck),
                         ("exchange", exchange), ("flags", flags), ("space", getattr(sub, "space", 0))):
            cases[name + "/" + key] = np.asarray(val)
        if env:
            cases[name + "/env"] = np.asarray(env)

    add("full_P2", models.mbl(16), Full(L=16), 2, 0, 10)
    add("full_P8", models.mbl(17), Full(L=17), 8, 0, 10)
    add("full_P3_window", models.mbl(12), Full(L=12), 3, 0, 0)
    add("sc_ref_P3", models.mbl(14), SpinConserve(14, 7), 3, 3, 0)
    add("sc3_P2", models.mbl(15), SpinConserve(15, 7), 2, 3, 6 | (4 << 8))
    add("sc3_P3", models.heisenberg(16), SpinConserve(16, 8), 3, 3, 6 | (4 << 8))
    # the transposed exchange, split and scheduled natively (dnm_mat_set_exchange)
    TR = _lib.EXCHANGE_TRANSPOSE
    add("tr_P4", models.mbl(17), Full(L=17), 4, 0, 0, exchange=TR)
    add("tr_P8_swz", models.heisenberg(19), Full(L=19), 8, 0, 6, exchange=TR)
    add("tr_P2", models.ising(16), Full(L=16), 2, 0, 0, exchange=TR)
    add("tr_P4_whole", models.long_range(17), Full(L=17), 4, 0, 0, exchange=TR, env="DNM_TRANSPOSE_PIPE=0")
    add("tr_P4_parity", models.mbl(18), Parity('odd', L=18), 4, 1, 0, exchange=TR)
    add("tr_P4_packed", models.heisenberg(18), Full(L=18), 4, 0, 0, exchange=TR, flags=_lib.MAT_REAL_PACKED)
    fn = os.path.join(str(tmp_path), "cases.npz")
    np.savez(fn, **cases)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native_comm_child.py"), fn], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and lines, "rc %d\n%s\n%s" % (out.returncode, out.stdout[-1500:], out.stderr[-3000:])
    rep = json.loads(lines[-1])
    assert rep["worst_relative"] < 1e-12 and rep.get("hooks") == "ok", rep
    assert "tiled=1" in rep["cases"]["full_P8"]["plan"] and "internal layout" in rep["cases"]["sc3_P3"]["plan"], rep
    assert {"tr_P4", "tr_P8_swz", "tr_P2", "tr_P4_whole", "tr_P4_parity", "tr_P4_packed"} <= set(rep["cases"]), rep


@pytest.mark.parametrize("world", [2, 3, 4])
def test_c_abi_alone_between_rank_processes(tmp_path, world):
    """Rank PROCESSES that bind the C ABI with ctypes alone (tests/native_ranks_child.py: no torch, no backend.py, no
    torch.distributed -- what a C / Cython / MPI host of include/dynamite_amd.h does): the communicator id through a file,
    dnm_comm_create with real ranks (stand-in transport), dnm_mat_mult_partitioned against the oracle for partner blocks,
    the transposed exchange (pipelined, Parity, real-packed) and column windows; dnm_comm_allreduce over the ranks; and the
    Krylov drivers ACROSS the processes through dnm_comm_hooks -- dnm_eigsolve against a dense solve, dnm_expm_multiply
    against scipy.  Replaces bpetsc_template_2.c:413-504, 787-879 and computations.py:89-112, 208-257."""
    import json
    import signal
    import subprocess
    import scipy.sparse.linalg as spla
    from dynamite_amd import models, _lib
    from dynamite_amd.subspaces import Full, SpinConserve, Parity
    from gpu_util import marshal, orc_msc, orc_sub, rand_state
    from oracle import oracle as orc
    cases = {}

    def add(name, H, sub, P, typ, swz, exchange=0, flags=0, solver=False):
        masks, offs, signs, coeffs = marshal(H)
        x = rand_state(sub.get_dimension(), seed=len(cases) + 1)
        if flags & _lib.MAT_REAL_PACKED:
            x = x.real + 0j
        y = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), x)
        if flags & _lib.MAT_REAL_PACKED:            # real vectors travel two amplitudes to an element
            x, y = x.real[0::2] + 1j * x.real[1::2], y.real[0::2] + 1j * y.real[1::2]
        L = sub.L
        nck = sub._nchoosek if typ == 3 else np.zeros((1, L + 1), dtype=np.int64)
        for key, val in (("masks", masks), ("mask_offsets", offs), ("signs", signs), ("coeffs", coeffs), ("x", x), ("y", y),
                         ("type", typ), ("L", L), ("k", getattr(sub, "k", 0)), ("P", P), ("swz", swz), ("nck", nck),
                         ("exchange", exchange), ("flags", flags), ("space", getattr(sub, "space", 0))):
            cases[name + "/" + key] = np.asarray(val)
        if solver:
            H.add_subspace(sub)
            Hs = H.to_numpy(subspaces=(sub, sub), sparse=True)
            cases[name + "/E0"] = np.asarray(spla.eigsh(Hs, k=1, which='SA', tol=1e-12, return_eigenvectors=False)[0])
            cases[name + "/z"] = spla.expm_multiply(-0.3j * Hs, x)
    TR = _lib.EXCHANGE_TRANSPOSE
    if world == 2:
        add("full_P2", models.mbl(16), Full(L=16), 2, 0, 10, solver=True)
        add("tr_P2", models.ising(16), Full(L=16), 2, 0, 0, exchange=TR)
        add("sc3_P2", models.mbl(15), SpinConserve(15, 7), 2, 3, 6 | (4 << 8))
        solver_case = "full_P2"
    elif world == 3:
        add("full_P3_window", models.mbl(12), Full(L=12), 3, 0, 0)
        add("sc_ref_P3", models.mbl(14), SpinConserve(14, 7), 3, 3, 0)
        add("sc3_P3", models.heisenberg(16), SpinConserve(16, 8), 3, 3, 6 | (4 << 8), solver=True)
        solver_case = "sc3_P3"
    else:
        add("tr_P4", models.mbl(17), Full(L=17), 4, 0, 0, exchange=TR, solver=True)
        add("tr_P4_parity", models.mbl(18), Parity('odd', L=18), 4, 1, 0, exchange=TR)
        add("tr_P4_packed", models.heisenberg(18), Full(L=18), 4, 0, 0, exchange=TR, flags=_lib.MAT_REAL_PACKED)
        add("full_P4_partner", models.mbl(16), Full(L=16), 4, 0, 6)
        solver_case = "tr_P4"
    fn = os.path.join(str(tmp_path), "cases.npz")
    np.savez(fn, **cases)
    env = dict(os.environ, DNM_RCCL_LIB=build_fake_rccl(), DNM_NATIVE_SOLVER_CASE=solver_case, DNM_FAKE_RCCL_TIMEOUT_S="300")
    env.setdefault("GPU_MAX_HW_QUEUES", "2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    idfile = os.path.join(str(tmp_path), "comm_id")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "native_ranks_child.py"), fn, str(r), str(world),
                               idfile, os.path.join(str(tmp_path), "rep%d.json" % r)], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
             for r in range(world)]
    errs = []
    try:
        for p in procs:
            errs.append(p.communicate(timeout=600)[1])
    finally:
        for p in procs:
            if p.poll() is None:
                os.killpg(p.pid, signal.SIGKILL)
    assert [p.returncode for p in procs] == [0] * world, "\n".join(e[-1500:] for e in errs)
    reps = [json.load(open(os.path.join(str(tmp_path), "rep%d.json" % r))) for r in range(world)]
    assert all(r["worst_relative"] < 1e-12 and len(r["cases"]) == len({k.split("/")[0] for k in cases}) for r in reps), reps
    assert all("eigsolve" in r and "expm" in r and r["eigsolve"]["matvecs"] > 5 for r in reps), reps
    assert len({r["eigsolve"]["E0"] for r in reps}) == 1          # every rank returns the same number, bit for bit


def test_native_exchange_runs_under_the_kernels():
    """The exchange of the native partitioned multiply must RUN AT THE SAME TIME as the rank-local kernels: on this system
    two streams of one priority can share a hardware queue and then serialise (round 6: every multi-rank time before was
    taken without any overlap), so the library's exchange stream has the highest priority (csrc/comm.cpp).  Rank 0 of 2 at
    L=30 with its exchange looped back over the real RCCL (tools/rccl_loopback_bench.py): whole multiply against messages
    alone + kernels alone (dnm_comm_set_phase) -- a good part of the shorter of the two must be hidden, even though in
    loop-back both compete for the same HBM."""
    import re
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DNM_COMM_PRIORITY"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_loopback_bench.py"), "30", "2", "0"], env=env,
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"one native call .*?: ([\d.]+) ms; its messages alone ([\d.]+) ms, its kernels alone ([\d.]+) ms: "
                  r"(-?[\d.]+) ms hidden \((-?\d+) % of the shorter\)", out.stdout)
    assert m, out.stdout[-2000:]
    whole, msgs, kern, hidden, pct = (float(v) for v in m.groups())
    assert whole < 0.93 * (msgs + kern) and pct >= 25, out.stdout[-1500:]
