"""What a rank of a partitioned SpinConserve multiply reads of its column window (BASELINE config 5 by default:
L=36, k=18, rank R of 8), on one GPU: window, needed ranges, time of the sweeps.

    python tools/window_needs.py [L=36] [k=18] [P=8]
"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dynamite_amd import backend, models
from dynamite_amd.subspaces import SpinConserve
from dynamite_amd.config import config
config.sc_layout = None      # this tool measures the reference-order kernels (tools/sc3_config5.py: the internal layout)
from gpu_util import marshal

L = int(sys.argv[1]) if len(sys.argv) > 1 else 36
k = int(sys.argv[2]) if len(sys.argv) > 2 else L // 2
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sub = SpinConserve(L, k)
dim = sub.get_dimension()
arrs = marshal(models.heisenberg(L))
for R in range(P):
    h = backend.create_mat(*arrs, sub._c(), sub._c(), flags=0, rank=R, nranks=P)
    mat = backend.ShellMat(h, sub._c(), sub._c(), P, R)
    t0 = time.perf_counter()
    lo, hi = mat.column_window()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    needs = mat.column_needs((lo, hi))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = mat.m_local
    own = (mat.row0, mat.row0 + n)
    remote = sum(b - a for a, b in needs) - n
    print("rank %d of %d: block %.2f G rows, window %.2f blocks, needed %.2f blocks in %d ranges -> receives %.1f GiB "
          "instead of %.1f GiB  (sweeps: window %.2f s, ranges %.2f s)" % (
              R, P, n / 1e9, (hi - lo + 1) / n, sum(b - a for a, b in needs) / n, len(needs), 16 * remote / 2**30,
              16 * (hi - lo + 1 - n) / 2**30, t1 - t0, t2 - t1), flush=True)
    mat.destroy()
