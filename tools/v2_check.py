#!/usr/bin/env python
"""GPU box: tile_pass2_kernel against tile_pass_kernel, with and without the XOR-swizzled layout.
For every (L, model) the reference is kernel 1 in natural order; every other configuration multiplies the
same logical vector (permuted into its layout) and must give the same logical result.
usage: python tools/v2_check.py [L ...]"""
import ctypes as C
import json
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import models, backend, msc_tools, _lib  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402


def phys_index(n, S, dev):
    i = torch.arange(n, dtype=torch.int64, device=dev)
    if not S:
        return i
    return i ^ (((i >> S) & ((1 << (S - 4)) - 1)) << 4)


def main():
    Ls = [int(a) for a in sys.argv[1:]] or [20, 24]
    config._initialize()
    cfgs = [dict(DNM_PLAN_MODE=0, DNM_KERNEL=1), dict(DNM_KERNEL=1), dict(DNM_KERNEL=2), dict(DNM_KERNEL=1, DNM_LOG_ROWS=4),
            dict(DNM_KERNEL=1, DNM_AMIN=4, DNM_GBITS=4), dict(DNM_KERNEL=2, DNM_AMIN=4, DNM_GBITS=4),
            dict(DNM_KERNEL=1, DNM_TILE_BITS=10, DNM_LOG_ROWS=2, DNM_PLAN_MODE=0),
            dict(DNM_KERNEL=1, DNM_TILE_BITS=11, DNM_LOG_ROWS=3, DNM_PLAN_MODE=1)]
    worst = 0.0
    for L in Ls:
        for name in ("mbl", "xxz", "ising", "long_range", "syk", "xsum"):
            if name in ("syk", "long_range") and L > 20:
                continue
            H = models.BY_NAME[name](L)
            H.establish_L()
            H.reduce_msc()
            masks, offs = msc_tools.get_mask_offsets(H.msc)
            sub = Full(L=L)
            dim = 1 << L
            x = backend.Vec(dim)          # index order: the logical vector
            x.set_random(3)
            xn = x.array.clone()
            ref = None
            for c in cfgs:
                for k in list(os.environ):
                    if k.startswith("DNM_") and k != "DNM_SWZ":
                        os.environ.pop(k)
                for k, v in c.items():
                    os.environ[k] = str(v)
                S = config.vec_swizzle
                mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c())
                p = phys_index(dim, S, xn.device)
                xs = backend.Vec(dim, swz=S)
                xs.array[p] = xn                  # element i lives at p[i]
                y = backend.Vec(dim, swz=S)
                mat.mult(xs, y)
                torch.cuda.synchronize()
                yl = y.array[p]
                if ref is None:
                    ref = yl.clone()
                err = float((yl - ref).abs().max())
                worst = max(worst, err)
                print("L=%d %-10s %-90s err=%.2e %s" % (L, name, json.dumps(c, separators=(',', ':')), err,
                                                         "" if err < 1e-10 else "  <<<<<< MISMATCH"), flush=True)
                mat.destroy()
    print("worst", worst)
    sys.exit(0 if worst < 1e-10 else 1)


if __name__ == "__main__":
    main()
