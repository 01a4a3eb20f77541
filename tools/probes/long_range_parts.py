#!/usr/bin/env python
"""GPU box: what each part of benchmark.py's long_range operator costs in the Full-space multiply at L=28 (round 6): the whole
operator, without its all-to-all ZZ terms (406 diagonal terms, 192 of them with one spin inside the LDS tile and one outside),
without its single flips, the ZZ terms alone."""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.operators import sigmax, sigmay, sigmaz, index_sum, op_sum  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
alpha = 1.13
xx = index_sum(sigmax(0) * sigmax(1), size=L)
zz = op_sum(index_sum(1 / (i ** alpha) * sigmaz(0) * sigmaz(i), size=L) for i in range(1, L))
fl = index_sum(0.5 * sigmax(), L) + index_sum(0.3 * sigmay(), L)
z1 = index_sum(0.1 * sigmaz(), L)
config._initialize()
for name, H in (("whole", xx + zz + fl + z1), ("without the all-to-all ZZ", xx + fl + z1), ("without the single flips", xx + zz + z1),
                ("ZZ + Z alone (diagonal)", zz + z1), ("XX bonds + single flips + nearest ZZ",
                                                        xx + fl + z1 + index_sum(sigmaz(0) * sigmaz(1), size=L))):
    H.L = L
    sub = Full(L=L)
    H.add_subspace(sub)
    mat = H.get_mat(subspaces=(sub, sub))
    xv, yv = mat.createVecs()
    xv.set_random(0)
    for _ in range(2):
        mat.mult(xv, yv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        mat.mult(xv, yv)
    e1.record()
    torch.cuda.synchronize()
    print("%-45s %7.3f ms   %s" % (name, e0.elapsed_time(e1) / 5, mat.describe().strip().replace("\n", " | ")[:230]), flush=True)
    del xv, yv, mat
    H.destroy_mat()
