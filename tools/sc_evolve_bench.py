#!/usr/bin/env python
"""evolve in SpinConserve(L, L/2) (Heisenberg chain): pure Krylov, default (hand-over) and Chebyshev."""
import os, sys, time
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.subspaces import SpinConserve  # noqa: E402
from dynamite_amd.computations import evolve  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
config._initialize()
sub = SpinConserve(L, L // 2)
H = models.heisenberg(L)
H.add_subspace(sub)
x = State(subspace=sub, state='random', seed=0)
ref = None
for algo in ('chebyshev', None, 'krylov'):
    y = State(subspace=sub)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        H.evolve(x, t=t, algo=algo, result=y)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("SpinConserve(%d,%d) evolve t=%g %-9s %s: %.3f s, %d multiplies" %
              (L, L // 2, t, algo or 'default', "first" if rep == 0 else "again", dt, evolve.last_stats['matvecs']),
              flush=True)
    if ref is None:
        ref = y
    else:
        d = y.copy()
        d.axpy(-1.0, ref)
        print("   |%s - chebyshev| = %.2e   norm %.12f" % (algo or 'default', d.norm(), y.norm()), flush=True)
