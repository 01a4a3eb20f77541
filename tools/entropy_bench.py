#!/usr/bin/env python
"""entanglement_entropy(state, keep = the k lowest spins) end to end: RDM kernel + dense Hermitian spectrum on the
device, against the reference's route (copy the matrix to the host, numpy eigvalsh) when asked for.
   entropy_bench.py [L] [k] [host] [sc]        (sc: a random state of SpinConserve(L, L/2))"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import computations as cp  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 26
k = int(sys.argv[2]) if len(sys.argv) > 2 else L // 2
config._initialize()
if "sc" in sys.argv[3:]:
    from dynamite_amd.subspaces import SpinConserve
    sub = SpinConserve(L, L // 2)
else:
    sub = Full(L=L)
st = State(L=L, subspace=sub)
st.set_random(seed=0, device_rng=True)
keep = list(range(k))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = cp.entanglement_entropy(st, keep)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("L=%d keep %d spins: entanglement_entropy %.3f s  (S = %.12f, Page value ~ %.6f)"
          % (L, k, dt, s, k * 0.6931471805599453 - 0.5 * 2.0 ** (2 * k - L)), flush=True)
if "host" in sys.argv[3:]:
    t0 = time.perf_counter()
    dm = cp.reduced_density_matrix(st, keep)
    t1 = time.perf_counter()
    s2 = cp.dm_entanglement_entropy(dm)
    t2 = time.perf_counter()
    print("   reference route: matrix to the host %.3f s + numpy eigvalsh %.2f s  (S = %.12f, |dS| = %.1e)"
          % (t1 - t0, t2 - t1, s2, abs(s2 - s)), flush=True)
