#!/usr/bin/env python
"""Multiply in an Explicit subspace (the states of SpinConserve(L, L/2) listed explicitly): what Auto / Explicit
subspaces cost against the dedicated SpinConserve kernel."""
import os, sys, time
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dynamite_amd import models, backend, msc_tools  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import SpinConserve, Explicit  # noqa: E402

config._initialize()
for L in [int(a) for a in sys.argv[1:]] or [24, 28]:
    sc = SpinConserve(L, L // 2)
    dim = sc.get_dimension()
    t0 = time.perf_counter()
    states = sc.idx_to_state(np.arange(dim, dtype=np.int64))
    sub = Explicit(states, L=L)
    t1 = time.perf_counter()
    H = models.mbl(L)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c())
    mat.precompute_diagonal()
    x, y = backend.Vec(dim), backend.Vec(dim)
    x.set_random(0)
    for _ in range(2):
        mat.mult(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        mat.mult(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("Explicit (states of SpinConserve(%d,%d)) dim=%d: %.3f ms  %.2f Gamp/s  (subspace built in %.1f s)  [%s]" %
          (L, L // 2, dim, ms, dim / ms / 1e6, t1 - t0, mat.describe().strip()), flush=True)
    mat.destroy()
