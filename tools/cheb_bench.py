#!/usr/bin/env python
"""evolve(): the default Krylov (Expokit-style Lanczos) against algo='chebyshev' -- time, multiplies, agreement.
usage: cheb_bench.py L [model] [t ...]"""
import os, sys, time
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.computations import evolve  # noqa: E402

L = int(sys.argv[1])
model = sys.argv[2] if len(sys.argv) > 2 else "mbl"
ts = [float(a) for a in sys.argv[3:]] or [0.2, 1.0]
config.L = L
config._initialize()
H = models.BY_NAME[model](L)
x = State(state='random', seed=0)
nrm = H.infinity_norm()
print("L=%d %s  ||H||_inf = %.3f" % (L, model, nrm), flush=True)
for t in ts:
    out = {}
    for algo in ('chebyshev', None):
        y = State()
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            H.evolve(x, t=t, algo=algo, result=y)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("  t=%-5g %-10s %s: %.3f s, %d multiplies, %d steps" %
                  (t, algo or 'default', "first" if rep == 0 else "again", dt, evolve.last_stats['matvecs'],
                   evolve.last_stats['its']), flush=True)
        out[algo] = y
    d = out[None].copy()
    d.axpy(-1.0, out['chebyshev'])
    print("  t=%-5g |default - chebyshev| = %.2e" % (t, d.norm()), flush=True)
