#!/usr/bin/env python
"""Time of large device allocations (hipMalloc / hipFree): single slabs against many pieces."""
import os, sys, time, ctypes as C
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import _lib  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
config._initialize()
L = _lib.lib()


def alloc(gbs):
    ps = []
    t0 = time.perf_counter()
    for gb in gbs:
        p = C.c_void_p()
        _lib.check(L.dnm_malloc(C.byref(p), C.c_size_t(int(gb * (1 << 30)))))
        ps.append(p)
    t1 = time.perf_counter()
    for p in ps:
        _lib.check(L.dnm_free(p))
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1


for name, gbs in (("1 x 40", [40]), ("1 x 64", [64]), ("1 x 80", [80]), ("1 x 120", [120]), ("1 x 160", [160]),
                  ("16 x 10", [10] * 16), ("4 x 40", [40] * 4), ("2 x 80", [80] * 2), ("1 x 160 again", [160]),
                  ("17 x 9.6", [9.6] * 17)):
    a, f = alloc(gbs)
    print("hipMalloc %-14s GiB: %.3f s, free %.3f s" % (name, a, f), flush=True)
