#!/usr/bin/env python
"""Multiply in XParity(SpinConserve(L, L/2)) (Heisenberg chain): the reduced operator carries one complemented
many-spin mask next to the chain bonds."""
import os, sys
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.subspaces import SpinConserve, XParity  # noqa: E402

config._initialize()
for L in [int(a) for a in sys.argv[1:]] or [28, 32]:
    sub = XParity(SpinConserve(L, L // 2), sector=+1)
    H = models.heisenberg(L)
    H.add_subspace(sub)
    x = State(subspace=sub, state='random', seed=1)
    y = State(subspace=sub)
    mat = H.get_mat()
    for _ in range(2):
        mat.mult(x.vec, y.vec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        mat.mult(x.vec, y.vec)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    dim = sub.get_dimension()
    print("XParity(SpinConserve(%d,%d)) dim=%d: %.3f ms  %.2f Gamp/s   [%s]" %
          (L, L // 2, dim, ms, dim / ms / 1e6, mat.describe().strip()), flush=True)
