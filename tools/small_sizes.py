#!/usr/bin/env python
"""Per-call wall time of the public entry points at SMALL sizes (launch / host overhead regime): small_sizes.py [L ...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402
from dynamite_amd.computations import evolve, eigsolve  # noqa: E402

Ls = [int(a) for a in sys.argv[1:]] or [10, 14, 18, 22]
config._initialize()


def wall(f, reps):
    f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for L in Ls:
    config.L = L
    sub = Full(L=L)
    H = models.mbl(L)
    H.add_subspace(sub)
    x = State(L=L, subspace=sub, state='random', seed=1)
    y = State(L=L, subspace=sub)
    reps = 20 if L <= 18 else 5
    t_dot = wall(lambda: H.dot(x, result=y), reps * 5)
    t_ev = wall(lambda: H.evolve(x, t=1.0, result=y), reps)
    mv_ev = evolve.last_stats['matvecs']
    t_ev01 = wall(lambda: H.evolve(x, t=0.1, result=y), reps)
    t_eig = wall(lambda: H.eigsolve(nev=1), max(2, reps // 4))
    mv_eig = eigsolve.last_stats['matvecs']
    t_ent = wall(lambda: x.entanglement_entropy(list(range(L // 2))), reps)
    t_norm = wall(lambda: x.norm(), reps * 5)
    print("L=%2d  dot %.3f ms | evolve(t=1) %.2f ms (%d mult) | evolve(t=0.1) %.2f ms | eigsolve(nev=1) %.1f ms (%d mult) | "
          "entanglement_entropy(L/2) %.2f ms | norm %.3f ms" % (L, t_dot, t_ev, mv_ev, t_ev01, t_eig, mv_eig, t_ent, t_norm), flush=True)
    H.destroy_mat()
