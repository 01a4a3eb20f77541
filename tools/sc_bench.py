#!/usr/bin/env python
"""Times the multiply in SpinConserve subspaces: sc_bench.py [--model NAME] L ...  (DNM_SC_LAYOUT=0: reference order)."""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models, backend, msc_tools  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import SpinConserve  # noqa: E402


def main():
    config._initialize()
    model = "mbl"
    args = list(sys.argv[1:])
    if "--model" in args:
        i = args.index("--model")
        model = args[i + 1]
        del args[i:i + 2]
    real = "--real" in args          # real arithmetic: the DNM_MAT_REAL_PACKED handle on double vectors
    args = [a for a in args if a != "--real"]
    for L in [int(a) for a in args] or [24, 28, 32]:
        k = L // 2
        H = models.BY_NAME[model](L)
        H.establish_L()
        H.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        sub = SpinConserve(L, k)
        dim = sub.get_dimension()
        mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c())
        x, y = mat.createVecs()
        print(mat.describe().strip(), flush=True)
        x.set_random(0)
        if real:
            import ctypes as C
            from dynamite_amd import _lib
            rmat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c(),
                                     flags=_lib.MAT_REAL_PACKED)
            xd = x.array.real.contiguous()
            yd = torch.zeros_like(xd)
            xp, yp = C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr())
            for _ in range(2):
                _lib.check(_lib.lib().dnm_mat_mult(rmat.handle, xp, yp, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _lib.check(_lib.lib().dnm_mat_mult(rmat.handle, xp, yp, None))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("SpinConserve L=%d k=%d dim=%d REAL arithmetic: %.3f ms  %.2f Grows/s  %.1f GB/s(16B)" %
                  (L, k, dim, ms, dim / ms / 1e6, 16.0 * dim / ms / 1e6), flush=True)
            rmat.destroy()
            del xd, yd
        for diag in (False, True):
            if diag:
                mat.precompute_diagonal()
            for _ in range(2):
                mat.mult(x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 5
            for _ in range(n):
                mat.mult(x, y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print("SpinConserve L=%d k=%d dim=%d diag_cached=%d: %.3f ms  %.2f Gamp/s  %.1f GB/s(32B)" %
                  (L, k, dim, diag, ms, dim / ms / 1e6, 32.0 * dim / ms / 1e6), flush=True)
        t0 = time.perf_counter()
        nrm = mat.norm()
        print("   norm %.6f in %.3f s" % (nrm, time.perf_counter() - t0), flush=True)
        mat.destroy()


if __name__ == "__main__":
    main()
