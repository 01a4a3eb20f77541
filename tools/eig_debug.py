#!/usr/bin/env python
"""eigsolve on a named model: prints every returned pair with its explicitly computed residual, norm and
overlaps (usage: eig_debug.py MODEL L NEV [TOL]); with DNM_KRYLOV_DEBUG=1 the solver adds its statistics."""
import os, sys
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dynamite_amd import models
name, L, nev = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-12
H = models.BY_NAME[name](L)
ev, vecs = H.eigsolve(nev=nev, which='lowest', tol=tol, getvecs=True)
A = H.to_numpy().toarray() if L <= 12 else None
for i, (e, v) in enumerate(zip(ev, vecs)):
    Hv = H.dot(v); r = Hv.copy(); r.axpy(-e, v)
    ov = max([abs(v.dot(vecs[j])) for j in range(i)] or [0])
    print("ev %.12f  res %.2e  |v|-1 %.1e  maxoverlap %.1e" % (e, r.norm(), abs(v.norm() - 1), ov))
if A is not None:
    print("exact:", np.linalg.eigvalsh(A)[:nev + 2])
