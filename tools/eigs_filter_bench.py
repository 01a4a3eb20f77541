#!/usr/bin/env python
"""eigsolve(nev > 1) at memory-bound sizes: the Chebyshev-filtered thick-restart Lanczos (default from 2^22 local
amplitudes on) against the plain restarted scheme (DNM_EIGS_FILTER=0) -- time, multiplies, agreement, residuals.
usage: eigs_filter_bench.py L [model] [nev] [tol] [which]"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.computations import eigsolve  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    model = sys.argv[2] if len(sys.argv) > 2 else "mbl"
    nev = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-10
    which = sys.argv[5] if len(sys.argv) > 5 else "lowest"
    plain = "--no-plain" not in sys.argv
    config._initialize()
    H = models.BY_NAME[model](L)
    H.establish_L()
    res = {}
    for mode in (["1", "0"] if plain else ["1"]):
        os.environ["DNM_EIGS_FILTER"] = mode
        for rep in range(2 if mode == "1" else 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            vals, vecs = H.eigsolve(nev=nev, tol=tol, which=which, getvecs=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st = eigsolve.last_stats
            # the reference's acceptance bars (tests/integration/test_eigsolve.py:17-88): residual and orthogonality
            worst_res, worst_orth = 0.0, 0.0
            from dynamite_amd.states import State
            hv = State(L=L)
            for i, (ev, v) in enumerate(zip(vals[:nev], vecs[:nev])):
                H.dot(v, hv)
                hv.vec.axpby(-ev, 1.0, v.vec)
                worst_res = max(worst_res, hv.norm() / abs(ev))
                for j in range(i):
                    worst_orth = max(worst_orth, abs(v.dot(vecs[j])))
            print("L=%d %s nev=%d tol=%.0e which=%s filter=%s: %.3f s, %d restarts, %d matvecs, nconv %d, "
                  "|Hv/ev - v| max %.2e, |<vi,vj>| max %.2e, evals %s"
                  % (L, model, nev, tol, which, mode, dt, st['its'], st['matvecs'], st['nconv'], worst_res, worst_orth,
                     np.array2string(np.asarray(vals[:nev]), precision=10)), flush=True)
            res[mode] = np.asarray(vals[:nev])
            del vecs
    if plain:
        print("   max |difference of the eigenvalues| %.2e" % np.max(np.abs(res["1"] - res["0"])), flush=True)


if __name__ == "__main__":
    main()
