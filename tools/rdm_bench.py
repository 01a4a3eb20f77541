#!/usr/bin/env python
"""Times reduced_density_matrix (GPU) for several cuts; prints effective FMA rate and bytes/s.
rdm_bench.py [L] [k]: with k, only the cut keep = range(k) on the Full space (for kernel-level profiles)."""
import os, sys, time
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import backend  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.states import State  # noqa: E402
from dynamite_amd.subspaces import Full, SpinConserve  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    config._initialize()
    only = int(sys.argv[2]) if len(sys.argv) > 2 else None
    for sub in ((Full(L=L),) if only else (Full(L=L), SpinConserve(L, L // 2))):
        st = State(L=L, subspace=sub, state='random', seed=0, ) if sub.get_dimension() <= (1 << 26) else None
        if st is None:
            st = State(L=L, subspace=sub)
            st.set_random(seed=0, device_rng=True)
        for keep in ([0], [L // 2], [3, 20], [1, 2, 3], [0, 9, L - 1], list(range(4)), list(range(L - 6, L)), list(range(0, 16, 2)), list(range(10)),
                     list(range(L // 2))) if not only else (list(range(only)),):
            if len(keep) > 13:
                continue
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                backend.reduced_density_matrix(st.vec, sub._to_c(), keep)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            k = len(keep)
            macs = (1 << L) * (1 << k) / 2          # complex MACs, lower triangle
            print("%-28s keep=%-22s %9.3f ms  %7.2f TFLOP/s (8 flop per complex MAC)  %7.1f GB/s of x"
                  % (repr(sub), str(keep if k <= 6 else "%d spins from %d" % (k, keep[0])), dt * 1e3,
                     8 * macs / dt / 1e12, sub.get_dimension() * 16 / dt / 1e9), flush=True)


if __name__ == "__main__":
    main()
