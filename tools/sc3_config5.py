#!/usr/bin/env python
"""BASELINE config 5 (SpinConserve(36, 18) Heisenberg chain on 8 ranks) in the internal three-field layout:
what every rank owns and reads (host tables only, runs anywhere), and -- with a GPU -- one rank's share of the
multiply at full size: its window of x in device memory, the two tiled passes, time per multiply.
usage: sc3_config5.py [L k P] [--rank R] [--real] [--native-loopback] [--order 1]
--order 1: the layout whose T blocks lie in the order made for partitions, what eigsolve without
eigenvectors solves on."""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import ctypes as C
import math
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from dynamite_amd import _lib, backend, models, msc_tools  # noqa: E402
from dynamite_amd.subspaces import SpinConserve  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    L, k, P = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (36, 18, 8)
    rank = int(sys.argv[sys.argv.index("--rank") + 1]) if "--rank" in sys.argv else None
    a, w = 14, 10
    order = int(sys.argv[sys.argv.index("--order") + 1]) if "--order" in sys.argv else 0
    sub = SpinConserve(L, k)
    d = _lib.Subspace.from_buffer_copy(sub._c())
    d.vec_swizzle = a | (w << 8) | (order << 16)
    dim = math.comb(L, k)
    H = models.heisenberg(L)
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    nint = C.c_int64()
    _lib.check(_lib.lib().dnm_vec_layout_size(C.byref(d), C.byref(nint)))
    print("SpinConserve(%d,%d): dim %d, internal length %d (+%.3f%%), %d ranks" % (L, k, dim, nint.value,
                                                                                 100.0 * (nint.value - dim) / dim, P))
    tot_need = 0
    links = []
    for r in range(P):
        h = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], d, d, False, _lib.MAT_HOST_ONLY, r, P)
        istart, ilen, nstart, nlen = backend.layout_partition(d, P, r)
        lo, hi = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().dnm_mat_column_window(h, C.byref(lo), C.byref(hi), None))
        nr = C.c_int64()
        _lib.check(_lib.lib().dnm_mat_column_ranges(h, 0, None, C.byref(nr)))
        if 0 < nr.value <= 1024:        # exact runs of blocks (what the schedules exchange when they are few enough)
            rg = (C.c_int64 * (2 * nr.value))()
            _lib.check(_lib.lib().dnm_mat_column_ranges(h, nr.value, rg, C.byref(nr)))
            need = [(int(rg[2 * i]), int(rg[2 * i + 1])) for i in range(nr.value)]
        else:
            shift = max(0, int(hi.value - lo.value + 1).bit_length() - 11)
            n = (hi.value >> shift) - (lo.value >> shift) + 1
            cmap = np.zeros(n, dtype=np.uint8)
            _lib.check(_lib.lib().dnm_mat_column_chunks(h, shift, cmap.ctypes.data_as(C.POINTER(C.c_uint8)), n, None))
            need = backend.needed_ranges(cmap, shift, (lo.value, hi.value))
        remote = sum(max(0, min(b, istart) - a_) + max(0, b - max(a_, istart + ilen)) for a_, b in need)
        tot_need += remote
        # by source: xGMI is point to point, a rank's exchange lasts as long as its busiest link
        own = [backend.layout_partition(d, P, q)[:2] for q in range(P)]
        by = [sum(max(0, min(b, o0 + ol) - max(a_, o0)) for a_, b in need) if q != r else 0 for q, (o0, ol) in enumerate(own)]
        links.append(by)
        print("  rank %d: rows %d (%.2f GiB), window %.2f GiB, reads %.2f GiB from other ranks in %d ranges; by source (GiB): %s"
              % (r, nlen, 16 * ilen / 2 ** 30, 16 * (hi.value - lo.value + 1) / 2 ** 30, 16 * remote / 2 ** 30, len(need),
                 " ".join("%.1f" % (16 * v / 2 ** 30) if v else "-" for v in by)))
        _lib.check(_lib.lib().dnm_mat_destroy(h))
    print("  received per multiply, mean over the ranks: %.2f GiB" % (16 * tot_need / P / 2 ** 30))
    busiest = max(max(row) for row in links)
    both = max(links[r][q] + links[q][r] for r in range(P) for q in range(P))
    print("  busiest link, one direction: %.2f GiB; both directions: %.2f GiB; links in use: %d of %d"
          % (16 * busiest / 2 ** 30, 16 * both / 2 ** 30, sum(1 for r in range(P) for q in range(r) if links[r][q] or links[q][r]),
             P * (P - 1) // 2))
    if rank is None:
        return
    import torch
    from dynamite_amd.config import config
    config._initialize()
    real = "--real" in sys.argv       # real arithmetic: one double per position, every position of the ABI in pairs
    h = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], d, d, False, _lib.MAT_REAL_PACKED if real else 0,
                           rank, P)
    mat = backend.ShellMat(h, d, d, P, rank)
    mat.real_packed = real
    print(mat.describe().strip() + (" [REAL arithmetic: sizes below count pairs of positions]" if real else ""))
    lo, hi = mat.column_window()
    wlen = hi - lo + 1
    print("rank %d: window of %.1f GiB, rows %.1f GiB" % (rank, 16 * wlen / 2 ** 30, 16 * mat.m_local / 2 ** 30), flush=True)
    xw = torch.empty(wlen, dtype=torch.complex128, device=config.device)
    # any finite content does for the timing; normal deviates by chunks
    for c0 in range(0, wlen, 1 << 28):
        c1 = min(wlen, c0 + (1 << 28))
        _lib.check(_lib.lib().dnm_vec_set_random(C.c_void_p(xw[c0:c1].data_ptr()), c1 - c0, 7, c0, backend._stream()))
    y = torch.empty(mat.m_local, dtype=torch.complex128, device=config.device)
    for _ in range(2):
        _lib.check(_lib.lib().dnm_mat_mult_window(mat.handle, C.c_void_p(xw.data_ptr()), lo, wlen, C.c_void_p(y.data_ptr()),
                                                  backend._stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n):
        _lib.check(_lib.lib().dnm_mat_mult_window(mat.handle, C.c_void_p(xw.data_ptr()), lo, wlen, C.c_void_p(y.data_ptr()),
                                                  backend._stream()))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    rows = backend.layout_partition(d, P, rank)[3]
    print("rank %d of %d, SpinConserve(%d,%d)%s: %.2f ms per multiply for %d rows = %.2f Grows/s"
          % (rank, P, L, k, " REAL" if real else "", ms, rows, rows / ms / 1e6))
    assert torch.isfinite(torch.view_as_real(y)).all()
    # the order of a partitioned multiply: the lo pass first (it needs the rank's own rows only and runs under the
    # exchange), then the window pass adds what reaches other blocks
    own0 = C.c_int64()
    _lib.check(_lib.lib().dnm_mat_ownership(mat.handle, C.byref(own0), None))
    xl = xw[own0.value - lo: own0.value - lo + mat.m_local]
    L_ = _lib.lib()
    for timed in (False, True):
        if timed:
            torch.cuda.synchronize()
            e0.record()
        for _ in range(n if timed else 2):
            _lib.check(L_.dnm_mat_mult_window_local(mat.handle, C.c_void_p(xl.data_ptr()), C.c_void_p(y.data_ptr()), backend._stream()))
            if timed:
                pass
            _lib.check(L_.dnm_mat_mult_window_remote(mat.handle, C.c_void_p(xw.data_ptr()), lo, wlen, C.c_void_p(y.data_ptr()),
                                                     backend._stream()))
    e1.record()
    torch.cuda.synchronize()
    print("   split as a partitioned multiply runs it (lo pass writes y, window pass adds): %.2f ms" % (e0.elapsed_time(e1) / n))
    if "--native-loopback" in sys.argv:
        # round 6: the same rank through dnm_mat_mult_partitioned with its exchange looped back over the real RCCL (a
        # communicator of one rank standing for rank `rank` of P; every peer's block = one buffer of the largest block's
        # size: the bytes mean nothing, the messages, their sizes and the overlap with the lo pass are the production
        # ones), whole / messages alone / kernels alone
        import time
        del xw
        torch.cuda.empty_cache()
        comm = backend.native_comm()
        unit = 2 if real else 1
        blocks = [backend.layout_partition(d, P, q)[1] // unit for q in range(P)]
        peer = torch.randn(2 * max(blocks), dtype=torch.float64, device=config.device).view(torch.complex128)
        px = (C.c_void_p * P)(*[peer.data_ptr()] * P)
        _lib.check(_lib.lib().dnm_comm_loopback(comm, rank, P, px, None))
        xl = peer[:mat.n_local]

        def timed(ph, nrep=4):
            _lib.check(L_.dnm_comm_set_phase(comm, ph))
            for _ in range(2):
                _lib.check(L_.dnm_mat_mult_partitioned(mat.handle, comm, C.c_void_p(xl.data_ptr()), C.c_void_p(y.data_ptr()), backend._stream()))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nrep):
                _lib.check(L_.dnm_mat_mult_partitioned(mat.handle, comm, C.c_void_p(xl.data_ptr()), C.c_void_p(y.data_ptr()), backend._stream()))
            torch.cuda.synchronize()
            _lib.check(L_.dnm_comm_set_phase(comm, _lib.PHASE_ALL))
            return (time.perf_counter() - t0) / nrep * 1e3
        tw, te, tc = timed(_lib.PHASE_ALL), timed(_lib.PHASE_EXCHANGE), timed(_lib.PHASE_COMPUTE)
        summ = mat.exchange_summary() if False else None
        print("   native call, exchange looped back over RCCL: %.2f ms per multiply; its messages alone %.2f ms, its kernels alone "
              "%.2f ms: %.2f ms hidden (%.0f %% of the shorter)" % (tw, te, tc, te + tc - tw, 100 * (te + tc - tw) / min(te, tc)))
        _lib.check(L_.dnm_comm_forget(comm, mat.handle))


if __name__ == "__main__":
    main()
