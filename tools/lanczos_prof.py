#!/usr/bin/env python
"""eigsolve(nev=1) alone (the basis-free Lanczos at large sizes), twice, for a kernel-level profile:
   bash tools/prof_cmd.sh OUT python3 tools/lanczos_prof.py L [real|complex] [tol]"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.computations import eigsolve  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mode = sys.argv[2] if len(sys.argv) > 2 else "complex"
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
config._initialize()
config.eigs_real_arithmetic = mode == "real"
H = models.mbl(L)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev = H.eigsolve(nev=1, tol=tol)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = eigsolve.last_stats
    print("L=%d %s eigsolve nev=1 tol=%.0e: %.3f s, %d matvecs (%.2f ms per step), E0=%.10f, residual %.1e, real=%s"
          % (L, mode, tol, dt, st['matvecs'], dt / st['matvecs'] * 1e3, ev[0], st['max_rel_residual'], st['real_arithmetic']),
          flush=True)
