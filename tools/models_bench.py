#!/usr/bin/env python
"""One table over the Hamiltonians the reference's harness and flagship example run (benchmarking/benchmark.py:129-170,
examples/scripts/kagome/run_kagome.py): for each (model, subspace, size) the plan line, ms per multiply, Gamp/s and the
32 B/amp rate; `--eigs` adds the wall time of eigsolve(nev=2) for the kagome cases, `--real` the multiply in real
arithmetic (what eigsolve runs for real-symmetric operators).

    python tools/models_bench.py CASE ...       CASE = model:subspace:L[:k]   e.g. kagome30:sc, bench_long_range:full:28,
                                                 bench_long_range:sc:28, kagome30:scx (SpinConserve + XParity),
                                                 ising:fullx:28 (XParity on the Full space), heisenberg:parity:28
"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.subspaces import Full, SpinConserve, XParity  # noqa: E402
from dynamite_amd.computations import eigsolve  # noqa: E402


def build(case):
    parts = case.split(":")
    model, space = parts[0], parts[1]
    if model.startswith("kagome"):
        H = models.kagome(model[len("kagome"):])
        L = H.L
    else:
        L = int(parts[2])
        H = models.BY_NAME[model](L)
    k = int(parts[3]) if len(parts) > 3 else L // 2
    if space == "full":
        sub = Full(L=L)
    elif space == "sc":
        sub = SpinConserve(L, k)
    elif space == "scx":
        sub = XParity(SpinConserve(L, k), sector='+' if L % 4 == 0 else '-')
    elif space == "fullx":
        sub = XParity(Full(L=L), sector='+')
    elif space == "parity":
        from dynamite_amd.subspaces import Parity
        sub = Parity('even', L=L)
    else:
        raise SystemExit("subspace: full / fullx (XParity on the Full space) / parity / sc / scx")
    H.allow_projection = True
    H.add_subspace(sub)
    return H, sub, L


def main():
    config._initialize()
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    eigs = "--eigs" in sys.argv
    for case in args:
        t0 = time.perf_counter()
        H, sub, L = build(case)
        mat = H.get_mat(subspaces=(sub, sub))
        t_build = time.perf_counter() - t0
        dim = sub.get_dimension()
        # the multiply in the matrix's own vectors (what the solvers run) ...
        xv, yv = mat.createVecs()
        xv.set_random(0)

        def timed(fn, n=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        ms = timed(lambda: mat.mult(xv, yv))
        # ... and through Operator.dot on states of the subspace (converted on the way when the layouts differ)
        x = State(L=L, subspace=sub)
        x.set_random(seed=0)
        y = State(L=L, subspace=sub)
        ms_dot = timed(lambda: H.dot(x, result=y), n=3)
        print("CASE %s  L=%d dim=%d nmasks=%d nterms=%d build %.2f s" % (case, L, dim, len(set(H.msc['masks'].tolist())),
                                                                       H.msc.size, t_build), flush=True)
        print("   plan: " + mat.describe().strip().replace("\n", " | "), flush=True)
        print("   multiply %.3f ms  %.2f Gamp/s  %.1f GB/s (32 B/amp)  frac %.3f   [Operator.dot on states of the "
              "subspace: %.3f ms%s]" %
              (ms, dim / ms / 1e6, 32.0 * dim / ms / 1e6, 32.0 * dim / ms / 1e6 / 8000.0, ms_dot,
               ", site relabelling %s" % (list(mat.perm_left),) if getattr(mat, "perm_left", None) else ""), flush=True)
        del xv, yv
        if "--real" in sys.argv:
            # the multiply eigsolve runs for a real-symmetric operator (DNM_MAT_REAL_PACKED): 16 B per amplitude
            pm = H.get_real_packed_mat(sub)
            if pm is None:
                print("   real arithmetic: no packed form", flush=True)
            else:
                # (raw arrays of the packed operator's size -- for timing: padding positions hold numbers too)
                from dynamite_amd.backend import RawVec
                xr = RawVec(torch.randn(2 * pm.n_local, dtype=torch.float64, device=config.device).view(torch.complex128),
                            pm.swz_right)
                yr = RawVec(torch.zeros(pm.m_local, dtype=torch.complex128, device=config.device), pm.swz_left)
                xr.perm, yr.perm = pm.perm_right, pm.perm_left
                msr = timed(lambda: pm.mult(xr, yr))
                print("   multiply in real arithmetic %.3f ms  %.2f Gamp/s  %.1f GB/s (16 B/amp)  frac %.3f" %
                      (msr, dim / msr / 1e6, 16.0 * dim / msr / 1e6, 16.0 * dim / msr / 1e6 / 8000.0), flush=True)
                del xr, yr
        if eigs:
            del x, y
            t0 = time.perf_counter()
            ev = H.eigsolve(nev=2, subspace=sub)
            torch.cuda.synchronize()
            st = dict(eigsolve.last_stats)
            print("   eigsolve(nev=2): %.2f s  %d multiplies  E0=%.10f E1=%.10f  real_arithmetic=%s residual %.2e" %
                  (time.perf_counter() - t0, st["matvecs"], ev[0], ev[1], st.get("real_arithmetic"),
                   st["max_rel_residual"]), flush=True)
        H.destroy_mat()
        del H
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
