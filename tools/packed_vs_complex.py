#!/usr/bin/env python
"""The real-arithmetic multiply of an L-spin operator (2^(L-1) packed elements) against the complex multiply of the
(L-1)-spin operator (as many elements, the same plan shape): what the records that see the packed bit cost.
   packed_vs_complex.py [L]"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
config._initialize()


def timed(mat, reps=10, lanczos=False):
    import ctypes as C
    from dynamite_amd import _lib
    x, y = mat.createVecs()
    z, _ = mat.createVecs()
    x.set_random(1)
    z.set_random(2)
    d = (C.c_double * 3)()

    def one():
        if lanczos:     # the multiply of a Lanczos step: y = A x - beta z with the fused <x, y> and |y|^2
            _lib.check(_lib.lib().dnm_mat_mult_lanczos(mat.handle, x.ptr, y.ptr, z.ptr, 0.3, d, None))
        else:
            mat.mult(x, y)
    for _ in range(3):
        one()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        one()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rep in range(2):
    Hc = models.mbl(L - 1)
    sc = Full(L=L - 1)
    Hc.add_subspace(sc)
    mc = Hc.get_mat(subspaces=(sc, sc))
    tc, tcl = timed(mc), timed(mc, lanczos=True)
    print("complex  L=%d: %.3f ms, Lanczos form %.3f ms   %s" % (L - 1, tc, tcl, mc.describe().splitlines()[0]), flush=True)
    Hc.destroy_mat()
    Hr = models.mbl(L)
    sr = Full(L=L)
    Hr.add_subspace(sr)
    mr = Hr.get_real_packed_mat(sr)
    tr, trl = timed(mr), timed(mr, lanczos=True)
    print("packed   L=%d: %.3f ms (%+.1f %%), Lanczos form %.3f ms (%+.1f %%)   %s"
          % (L, tr, (tr / tc - 1) * 100, trl, (trl / tcl - 1) * 100, mr.describe().splitlines()[0]), flush=True)
    Hr.destroy_mat()
