#!/usr/bin/env python
"""Ablation probe (GPU box): times the tiled kernel skeleton on synthetic
operators with a controlled number of in-tile masks / window passes, next to a
plain device copy."""
import ctypes as C
import json
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from dynamite_amd import backend, msc_tools, _lib  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.operators import Operator, sigmax, sigmay, sigmaz  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402


def bonds_model(L, bonds, fields=True):
    terms = []
    for i in bonds:
        terms += [(3 << i, 0, 0.25), (3 << i, 3 << i, -0.25), (0, 3 << i, 0.25)]
    if fields:
        terms += [(0, 1 << i, 0.1 * (i + 1)) for i in range(L)]
    H = Operator(msc=terms)
    H.L = L
    return H


def timeit(fn, n=5):
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


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    config._initialize()
    dim = 1 << L
    sub = Full(L=L)
    x, y = backend.Vec(dim, swz=sub.vec_swizzle), backend.Vec(dim, swz=sub.vec_swizzle)
    x.set_random(0)
    ms = timeit(lambda: y.array.copy_(x.array))
    print("torch copy            %8.3f ms  %7.1f GB/s (r+w)" % (ms, 32.0 * dim / ms / 1e6), flush=True)
    ms = timeit(lambda: _lib.lib().dnm_vec_scale(y.ptr, dim, 1.0001, 0.0, None))
    print("vec_scale (rmw)       %8.3f ms  %7.1f GB/s (r+w)" % (ms, 32.0 * dim / ms / 1e6), flush=True)
    ms = timeit(lambda: _lib.lib().dnm_vec_axpby(y.ptr, x.ptr, dim, 0.5, 0.0, 0.5, 0.0, None))
    print("vec_axpby (2r+w)      %8.3f ms  %7.1f GB/s" % (ms, 48.0 * dim / ms / 1e6), flush=True)

    cases = []
    for B, R in ((12, 3), (12, 4), (13, 3), (13, 4), (11, 3)):
        for glds in (0,):
            for nb in (0, 1, 4, B - 1):
                cases.append(dict(B=B, R=R, glds=glds, bonds=list(range(nb)), tag="low%d" % nb))
    # accumulate passes on a window
    for B, R in ((12, 3), (13, 3)):
        for glds in (0,):
            for amin in (3, 4, 5, 6):
                w = 14
                cases.append(dict(B=B, R=R, glds=glds, amin=amin, bonds=list(range(w, w + B - amin - 1)),
                                  fields=False, tag="win@%d a>=%d" % (w, amin)))
    for c in cases:
        os.environ["DNM_TILE_BITS"] = str(c["B"])
        os.environ["DNM_LOG_ROWS"] = str(c["R"])
        os.environ["DNM_PLAN_MODE"] = "0"
        os.environ["DNM_AMIN"] = str(c.get("amin", 3))
        H = bonds_model(L, c["bonds"], c.get("fields", True))
        H.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        flags = _lib.MAT_USE_GLDS if c["glds"] else 0
        mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c(), flags=flags)
        nl = C.c_int()
        _lib.lib().dnm_mat_plan_launches(mat.handle, C.byref(nl))
        ms = timeit(lambda: mat.mult(x, y))
        print("B=%d R=%d glds=%d %-14s launches=%d %8.3f ms  (%.1f GB/s @32B/amp/launch-equiv)" % (
            c["B"], c["R"], c["glds"], c["tag"], nl.value, ms, 32.0 * dim / ms / 1e6), flush=True)
        if os.environ.get("PROBE_DESCRIBE"):
            print(mat.describe())
        mat.destroy()


if __name__ == "__main__":
    main()
