#!/usr/bin/env python
"""eigsolve(nev=1) on SpinConserve(L, L/2) in complex128 and in real arithmetic: sc_eigs_real.py [L] [tol] [real]
(a third argument `real` runs the real-arithmetic solve only, twice -- sizes whose complex vectors do not fit)"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import SpinConserve  # noqa: E402
from dynamite_amd.computations import eigsolve  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
config._initialize()
sub = SpinConserve(L, L // 2)
H = models.heisenberg(L)
H.add_subspace(sub)
for real in ((True, True) if len(sys.argv) > 3 and sys.argv[3] == 'real' else (False, True, False, True)):
    config.eigs_real_arithmetic = real
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev = H.eigsolve(nev=1, tol=tol, subspace=sub)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = eigsolve.last_stats
    print("SpinConserve(%d,%d) eigsolve nev=1 tol=%.0e %s: %.3f s, %d matvecs (%.2f ms per step), E0=%.10f, residual %.1e"
          % (L, L // 2, tol, "real   " if st['real_arithmetic'] else "complex", dt, st['matvecs'], dt / st['matvecs'] * 1e3,
             ev[0], st['max_rel_residual']), flush=True)
