#!/usr/bin/env python
"""Ground state of the Heisenberg chain in SpinConserve(L, L/2) on one GPU -- the solver of BASELINE config 5 (Lanczos
without a stored basis, internal layout) at the largest size one MI355X holds: sc_eigs_big.py [L] [tol].  Reports
the multiply time, the solve, and the residual |H v - lambda v| measured from the returned state."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.subspaces import SpinConserve  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 34
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
    config.L = L
    config._initialize()
    sub = SpinConserve(L, L // 2)
    H = models.heisenberg(L)
    H.add_subspace(sub)
    print("SpinConserve(%d,%d): %d states, %.1f GiB per vector, layout code %d" % (L, L // 2, sub.get_dimension(),
          16.0 * sub.get_dimension() / 2 ** 30, sub.vec_swizzle), flush=True)
    print(H.get_mat().describe().strip(), flush=True)
    x = State(subspace=sub, state='random', seed=1)
    y = State(subspace=sub)
    for _ in range(2):
        H.dot(x, result=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        H.dot(x, result=y)
    torch.cuda.synchronize()
    print("multiply: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
    del x, y
    t0 = time.perf_counter()
    ev, vecs = H.eigsolve(nev=1, tol=tol, getvecs=True, subspace=sub)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    v = vecs[0]
    w = H.dot(v)
    w.axpy(-ev[0], v)
    res = w.norm()
    print("eigsolve(nev=1, tol=%g): %.2f s, E0 = %.12f (E0/L = %.6f), |Hv - E0 v| = %.2e, |v| = %.12f, peak memory %.1f GiB"
          % (tol, t1, ev[0], ev[0] / L, res, v.norm(), torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
    assert res <= 10 * tol * max(1.0, abs(ev[0])), "residual above the tolerance"


if __name__ == "__main__":
    main()
