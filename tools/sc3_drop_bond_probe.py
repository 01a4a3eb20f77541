#!/usr/bin/env python
"""VERDICT r4 task 3, the bound before the build: what can taking the top T bond(s) of a SpinConserve chain into the
window pass's LDS tile save AT MOST?  The Heisenberg chain at SpinConserve(L, L/2) with its top 0 / 1 / 2 bonds removed
from the operator -- the passes then simply do not gather them.  A pair-of-T-patterns tile would still pay an LDS hop
for each, so the time saved here bounds its gain from above.  Per-pass times: run under tools/prof_cmd.sh.

    python tools/sc3_drop_bond_probe.py [L]
"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import backend, msc_tools  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum  # noqa: E402
from dynamite_amd.subspaces import SpinConserve  # noqa: E402


def chain(L, drop):
    bonds = [(i, i + 1) for i in range(L - 1) if i < L - 1 - drop]
    H = op_sum(op_sum(0.25 * s(i) * s(j) for s in (sigmax, sigmay, sigmaz)) for i, j in bonds)
    H.L = L
    return H


def main():
    config._initialize()
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    sub = SpinConserve(L, L // 2)
    for drop in ([int(os.environ['DROP'])] if 'DROP' in os.environ else (0, 1, 2, 3)):
        H = chain(L, drop)
        H.establish_L()
        H.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c(), site_perm=False)
        x, y = mat.createVecs()
        x.set_random(0)
        for _ in range(3):
            mat.mult(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            mat.mult(x, y)
        e1.record()
        torch.cuda.synchronize()
        print("top %d bond(s) dropped: %.3f ms per multiply   %s" % (drop, e0.elapsed_time(e1) / 10,
                                                                     mat.describe().strip()[:150]), flush=True)
        mat.destroy()
        del x, y


if __name__ == "__main__":
    main()
