#!/usr/bin/env python
"""The tiled multiply at sizes around the Infinity Cache (256 MB): streaming (non-temporal) y traffic against plain.
usage: python tools/policy_sizes.py L [L ...]   (DNM_CACHE_POLICY values: 226 default, 224 plain y loads, 162 plain y
stores, 160 both plain)"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from dynamite_amd import models, backend, msc_tools  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402


def main():
    config._initialize()
    for L in [int(a) for a in sys.argv[1:]]:
        sub = Full(L=L)
        dim = 1 << L
        x, y = backend.Vec(dim, swz=sub.vec_swizzle), backend.Vec(dim, swz=sub.vec_swizzle)
        x.set_random(0)
        H = models.BY_NAME["mbl"](L)
        H.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        for cp in os.environ.get("POLICIES", "226,224,162,160").split(","):
            os.environ["DNM_CACHE_POLICY"] = cp
            mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c())
            n = 200 if L <= 24 else 20
            for _ in range(5):
                mat.mult(x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                mat.mult(x, y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print("L=%d policy %-4s %9.4f ms  %7.2f Gamp/s" % (L, cp, ms, dim / ms / 1e6), flush=True)
            mat.destroy()
        del x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
