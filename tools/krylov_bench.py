#!/usr/bin/env python
"""Times Operator.evolve / Operator.eigsolve (BASELINE config 2: L=26 XXZ, Full space, one GPU)."""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.computations import evolve, eigsolve  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    model = sys.argv[2] if len(sys.argv) > 2 else "xxz"
    config._initialize()
    H = models.BY_NAME[model](L)
    psi = State(L=L, state='random', seed=0)
    t0 = time.perf_counter()
    nrm = H.infinity_norm()
    torch.cuda.synchronize()
    print("L=%d %s  infinity_norm=%.6f (%.3f s incl. build)" % (L, model, nrm, time.perf_counter() - t0), flush=True)
    out = State(L=L)
    y = State(L=L)
    for _ in range(2):
        H.dot(psi, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        H.dot(psi, y)
    torch.cuda.synchronize()
    tm = (time.perf_counter() - t0) / 10
    print("matvec %.3f ms" % (tm * 1e3), flush=True)
    for ncv in (30, 15):
        for t in (1.0, 10.0 / nrm):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            H.evolve(psi, t=t, result=out, ncv=ncv)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st = evolve.last_stats
            print("evolve t=%.4f ncv=%d: %.3f s, %d outer steps, %d matvecs (%.1f ms/matvec-equivalent, matvec share %.0f%%), |y|=%.12f"
                  % (t, ncv, dt, st['its'], st['matvecs'], dt / st['matvecs'] * 1e3, 100 * st['matvecs'] * tm / dt, out.norm()), flush=True)
    for nev in (1, 5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev = H.eigsolve(nev=nev, tol=1e-10)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = eigsolve.last_stats
        print("eigsolve nev=%d: %.3f s, %d restarts, %d matvecs (matvec share %.0f%%), E0=%.10f"
              % (nev, dt, st['its'], st['matvecs'], 100 * st['matvecs'] * tm / dt, ev[0]), flush=True)


if __name__ == "__main__":
    main()
