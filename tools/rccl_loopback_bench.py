#!/usr/bin/env python
"""The partitioned Full-space multiply (partner blocks) of ONE rank at full size on ONE GPU, with its exchange looped
back over RCCL: a process group of world size 1 on the nccl backend, every send and receive addressed to the rank
itself (tests/rccl_self_child.py stage E has the small, oracle-checked form).  What it measures: the production code
path of ShellMat.mult -- batch_isend_irecv on RCCL's stream, the rank-local passes under it, per-block waits, the partner
passes -- with full-size messages (GiB each) and a link that is as fast as device memory, i.e. the compute side of a
partitioned step and what RCCL's own copy kernels cost the passes they overlap with.  NOT an xGMI measurement.
    python tools/rccl_loopback_bench.py [L P rank] ...      (default: 31 2 0   33 8 5)"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import datetime
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    # (RCCL's stream with high priority, as bench.py creates its group: _comm.nccl_options; --default-priority: without)
    from dynamite_amd import _comm
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                            timeout=datetime.timedelta(seconds=300),
                            pg_options=None if "--default-priority" in sys.argv else _comm.nccl_options())
    from dynamite_amd import backend, models, msc_tools, _lib
    from dynamite_amd.config import config
    from dynamite_amd.subspaces import Full
    config._initialize()
    args = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [31, 2, 0, 33, 8, 5]
    for L, P, me in zip(args[0::3], args[1::3], args[2::3]):
        sub = Full(L=L)
        H = models.mbl(L)
        H.establish_L()
        H.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        c = sub._c()
        c.vec_swizzle = 16
        n_loc = (1 << L) // P
        h = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], c, c, False, 0, me, P)
        mat = backend.ShellMat(h, c, c, P, me)
        psends = {}
        for q in mat.partners:
            hq = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], c, c, False, _lib.MAT_HOST_ONLY, q, P)
            sq, _ = backend.exchange_plan(hq)
            psends[q] = [t for t in sq if t[0] == me]
            _lib.check(_lib.lib().dnm_mat_destroy(hq))
        taken = {q: 0 for q in mat.partners}
        loop = []
        for q, off, cnt in mat.recvs:
            _, soff, scnt = psends[q][taken[q]]
            taken[q] += 1
            loop.append((0, soff, scnt))
        recv_bytes = 16 * sum(cnt for _, _, cnt in mat.recvs)
        mat.sends = loop
        mat.recvs = [(0, off, cnt) for _, off, cnt in mat.recvs]
        x, y = backend.Vec(n_loc, swz=16), backend.Vec(n_loc, swz=16)
        x.set_random(1)
        print("L=%d rank %d of %d: block 2^%d amplitudes, receives %.1f GiB in %d blocks per multiply"
              % (L, me, P, n_loc.bit_length() - 1, recv_bytes / 2 ** 30, len(mat.recvs)), flush=True)
        print(mat.describe().strip(), flush=True)

        def timed(fn, n=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        t_full = timed(lambda: mat.mult(x, y))
        t_exch = timed(lambda: mat.exchange_only(x))
        # the same multiply as ONE native call (dnm_mat_mult_partitioned, csrc/comm.cpp): the library's own communicator
        # (sharing this process's RCCL) standing for this rank, every peer's block = this rank's x (round 5)
        import ctypes as C
        comm = backend.native_comm()
        px = (C.c_void_p * P)(*[x.array.data_ptr()] * P)
        _lib.check(_lib.lib().dnm_comm_loopback(comm, me, P, px, None))
        t_native = timed(lambda: _lib.check(_lib.lib().dnm_mat_mult_partitioned(mat.handle, comm, x.ptr, y.ptr, backend._stream())))
        def phases(handle, xv, yv):
            """the native call's messages alone / kernels alone (dnm_comm_set_phase, round 6)"""
            out = []
            for ph in (_lib.PHASE_EXCHANGE, _lib.PHASE_COMPUTE):
                _lib.check(_lib.lib().dnm_comm_set_phase(comm, ph))
                out.append(timed(lambda: _lib.check(_lib.lib().dnm_mat_mult_partitioned(handle, comm, xv.ptr, yv.ptr, backend._stream()))))
            _lib.check(_lib.lib().dnm_comm_set_phase(comm, _lib.PHASE_ALL))
            return out
        t_ne, t_nc = phases(mat.handle, x, y)
        _lib.check(_lib.lib().dnm_comm_forget(comm, mat.handle))
        print("   the same as one native call (dnm_mat_mult_partitioned, exchange on the library's own stream): %.2f ms; its messages "
              "alone %.2f ms, its kernels alone %.2f ms: %.2f ms hidden (%.0f %% of the shorter)"
              % (t_native, t_ne, t_nc, t_ne + t_nc - t_native, 100 * (t_ne + t_nc - t_native) / min(t_ne, t_nc)), flush=True)
        L_ = _lib.lib()
        t_local = timed(lambda: _lib.check(L_.dnm_mat_mult_local(mat.handle, x.ptr, y.ptr, backend._stream())))

        def remote_only():
            for i in range(len(mat.recvs)):
                _lib.check(L_.dnm_mat_mult_remote(mat.handle, i, backend.C.c_void_p(mat._recv[i].data_ptr()), y.ptr,
                                                  backend._stream()))
        t_remote = timed(remote_only)
        print("   multiply with the exchange looped back over RCCL: %.2f ms; exchange alone %.2f ms (%.0f GB/s through RCCL's "
              "copies); rank-local passes alone %.2f ms; partner passes alone %.2f ms; local + partner %.2f ms"
              % (t_full, t_exch, recv_bytes / t_exch / 1e6, t_local, t_remote, t_local + t_remote), flush=True)
        assert torch.isfinite(torch.view_as_real(y.array)).all()
        mat.destroy()
        del x, y
        torch.cuda.empty_cache()
        if P >= 4:
            # the transposed exchange (the default from four ranks on): its two all-to-alls looped back -- every piece
            # lands where the same piece of a peer would (the bytes are this rank's own, so the RESULT means nothing;
            # the schedule, the message sizes and the overlap of RCCL's stream with the passes are the production ones)
            p_ = P.bit_length() - 1
            S = 16
            cap = (L - 2 * p_ - 1 - 2 + 4) // 2
            if cap < S:
                S = 14 if cap == 15 else cap
            c2 = Full(L=L)._c()
            c2.vec_swizzle = S
            h2 = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], c2, c2, False, 0, me, P)
            mat2 = backend.ShellMat(h2, c2, c2, P, me)
            split = backend.transpose_split(masks, offs, H.msc['signs'], H.msc['coeffs'], L, P, S, 0)
            assert split is not None
            mat2.set_transposed(split, c2, c2, 0)
            lo_h, hi_h, pieces, own, cnt = mat2._tr
            mat2._tr = (lo_h, hi_h, [(0, off, c_) for _, off, c_ in pieces], own, cnt)
            x, y = backend.Vec(n_loc, swz=S), backend.Vec(n_loc, swz=S)
            x.set_random(1)
            t_tr = timed(lambda: mat2.mult(x, y))
            t_ex = timed(lambda: mat2.exchange_only(x))
            moved = 2 * 16 * cnt * len(pieces)
            print("   transposed exchange looped back (swizzle %d, %s): %.2f ms per multiply; its two all-to-alls alone %.2f ms "
                  "(%.1f GiB, %.0f GB/s through RCCL's copies)"
                  % (S, "pipelined" if mat2._tr_pipe else "whole pieces", t_tr, t_ex, moved / 2 ** 30, moved / t_ex / 1e6), flush=True)
            # ... and split and scheduled inside the library (dnm_mat_set_exchange + dnm_mat_mult_partitioned): the host's
            # buffers go first (the library has its own pair), every peer's block = this rank's x, no peer handles
            # (the returning pieces carry this rank's own result)
            for h_ in mat2._tr[:2]:
                _lib.check(_lib.lib().dnm_mat_destroy(h_))
            mat2._tr, mat2._tr_bufs = None, None
            torch.cuda.empty_cache()
            assert mat2.set_native_transposed()
            px = (C.c_void_p * P)(*[x.array.data_ptr()] * P)
            _lib.check(_lib.lib().dnm_comm_loopback(comm, me, P, px, None))
            t_ntr = timed(lambda: _lib.check(_lib.lib().dnm_mat_mult_partitioned(mat2.handle, comm, x.ptr, y.ptr, backend._stream())))
            t_ne, t_nc = phases(mat2.handle, x, y)
            _lib.check(_lib.lib().dnm_comm_forget(comm, mat2.handle))
            print("   the same as one native call (split by dnm_mat_set_exchange, schedule in csrc/comm.cpp): %.2f ms; its messages "
                  "alone %.2f ms, its kernels alone %.2f ms: %.2f ms hidden (%.0f %% of the shorter)"
                  % (t_ntr, t_ne, t_nc, t_ne + t_nc - t_ntr, 100 * (t_ne + t_nc - t_ntr) / min(t_ne, t_nc)), flush=True)
            assert torch.isfinite(torch.view_as_real(y.array)).all()
            mat2.destroy()
            del x, y
            torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
