#!/usr/bin/env python
"""Full-size consistency checks of the SpinConserve kernels (no oracle at these sizes):
  1. SpinConserve(34,17), 2.33 G rows on one GPU: block kernel against the row kernel, and Hermiticity;
  2. one rank of BASELINE config 5 (L=36, k=18 on 8 GPUs): rows of rank R through the column window,
     block kernel against the row kernel, with the time per multiply.
usage: sc_fullsize_check.py [single|rank R]"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models, backend, msc_tools, _lib  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import SpinConserve  # noqa: E402
config.sc_layout = None      # this tool measures the reference-order kernels (tools/sc3_config5.py: the internal layout)


def arrays(L):
    H = models.heisenberg(L) if hasattr(models, 'heisenberg') else models.mbl(L)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    return masks, offs, H.msc['signs'], H.msc['coeffs']


def timed(f, n=3):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def single():
    L, k = 34, 17
    sub = SpinConserve(L, k)
    dim = sub.get_dimension()
    arrs = arrays(L)
    mats = {}
    for blk in ("13", "0"):
        os.environ["DNM_SC_BLOCK"] = blk
        mats[blk] = backend.build_mat(*arrs, sub._to_c(), sub._to_c())
        mats[blk].precompute_diagonal()
        print(blk, mats[blk].describe().strip(), flush=True)
    a, b, Ha, Hb = (backend.Vec(dim) for _ in range(4))
    a.set_random(1); b.set_random(2)
    a.normalize(); b.normalize()
    tb = timed(lambda: mats["13"].mult(a, Ha))
    tr = timed(lambda: mats["0"].mult(a, Hb))
    print("SpinConserve(%d,%d) dim=%d: block %.2f ms (%.1f Gamp/s), row kernel %.2f ms" %
          (L, k, dim, tb, dim / tb / 1e6, tr), flush=True)
    Hb.axpby(-1.0, 1.0, Ha)
    print("   |block - row| = %.3e   |Ha| = %.6f" % (Hb.norm(), Ha.norm()), flush=True)
    mats["13"].mult(b, Hb)
    print("   Hermiticity |<b,Ha> - <Hb,a>| = %.3e" % abs(b.dot(Ha) - Hb.dot(a)), flush=True)


def one_rank(R, P=8):
    L, k = 36, 18
    sub = SpinConserve(L, k)
    dim = sub.get_dimension()
    arrs = arrays(L)
    Lb = _lib.lib()
    out = {}
    for blk in ("13", "0"):
        os.environ["DNM_SC_BLOCK"] = blk
        h = backend.create_mat(*arrs, sub._c(), sub._c(), flags=0, rank=R, nranks=P)
        mat = backend.ShellMat(h, sub._c(), sub._c(), P, R)
        mat.precompute_diagonal()
        lo, hi = mat.column_window()
        n = mat.m_local
        if blk == "13":
            print("rank %d of %d: rows [%d, %d), window [%d, %d] = %.2f x the block; %s" %
                  (R, P, mat.row0, mat.row0 + n, lo, hi, (hi - lo + 1) / n, mat.describe().strip()), flush=True)
            xw = backend.Vec(hi - lo + 1)
            xw.set_random(3)
        y = backend.Vec(n)
        t = timed(lambda: _lib.check(Lb.dnm_mat_mult_window(mat.handle, xw.ptr, lo, hi - lo + 1, y.ptr, None)))
        print("   DNM_SC_BLOCK=%s: %.2f ms per multiply (%.1f Gamp/s per GPU)" % (blk, t, n / t / 1e6), flush=True)
        out[blk] = y
        mat.destroy()
    out["0"].axpby(-1.0, 1.0, out["13"])
    print("   |block - row| = %.3e   |y| = %.6f" % (out["0"].norm(), out["13"].norm()), flush=True)


if __name__ == "__main__":
    config._initialize()
    if len(sys.argv) > 1 and sys.argv[1] == "rank":
        one_rank(int(sys.argv[2]))
    else:
        single()
