#!/usr/bin/env python
"""y = Hx with the default plan over a range of sizes (random-field Heisenberg, Full space)."""
import os, sys
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models, backend, msc_tools  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402

config._initialize()
for L in [int(a) for a in sys.argv[1:]] or list(range(16, 31, 2)):
    H = models.mbl(L)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    sub = Full(L=L)
    mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c())
    dim = 1 << L
    x, y = mat.createVecs()
    x.set_random(0)
    n = 20 if L >= 26 else 200
    for _ in range(3):
        mat.mult(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        mat.mult(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("L=%2d  %9.4f ms  %6.2f Gamp/s  %7.1f GB/s (32 B/amp)  frac %.3f   %s" %
          (L, ms, dim / ms / 1e6, 32.0 * dim / ms / 1e6, 32.0 * dim / ms / 1e6 / 8000.0,
           mat.describe().splitlines()[0]), flush=True)
    mat.destroy()
    del x, y
