#!/usr/bin/env python
"""Times y = Hx for a list of plan configurations in one process (GPU box).
usage: python tools/sweep.py L [model] -- prints one line per configuration."""
import ctypes as C
import itertools
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


def main():
    L = int(sys.argv[1])
    model = sys.argv[2] if len(sys.argv) > 2 else "mbl"
    cfgs = json.loads(os.environ.get("SWEEP", "null")) or [
        dict(B=12, R=3, mode=0, amin=3), dict(B=12, R=4, mode=0, amin=3), dict(B=12, R=3, mode=0, amin=3, glds=1),
        dict(B=13, R=3, mode=0, amin=3), dict(B=13, R=4, mode=0, amin=3), dict(B=13, R=3, mode=0, amin=4),
        dict(B=11, R=3, mode=0, amin=3), dict(B=12, R=3, mode=0, amin=4), dict(B=12, R=3, mode=0, amin=6),
        dict(B=12, R=3, mode=1, amin=3), dict(B=13, R=3, mode=1, amin=3), dict(B=13, R=4, mode=1, amin=3),
        dict(B=11, R=3, mode=1, amin=3), dict(gather=1),
    ]
    config._initialize()
    sub = Full(L=L)

    def operator_for(c):
        if "bonds" in c:     # synthetic: Heisenberg bonds on the listed sites only (+ fields)
            from dynamite_amd.operators import Operator
            terms = []
            for i in c["bonds"]:
                terms += [(3 << i, 0, 0.25), (3 << i, 3 << i, -0.25)]
                if not c.get("nodiag"):
                    terms += [(0, 3 << i, 0.25)]
            if not c.get("nodiag"):
                terms += [(0, 1 << i, 0.1 * (i + 1)) for i in range(L)]
            H = Operator(msc=terms)
            H.L = L
        else:
            H = models.BY_NAME[model](L)
        H.reduce_msc()
        return H
    dim = 1 << L
    x, y = backend.Vec(dim, swz=sub.vec_swizzle), backend.Vec(dim, swz=sub.vec_swizzle)
    x.set_random(0)
    x.normalize()
    ref = None
    for c in cfgs:
        flags = 0
        for k in list(os.environ):
            if k.startswith("DNM_") and k not in ("DNM_FUZZ_N", "DNM_SWZ"):
                os.environ.pop(k)
        for k, v in c.get("env", {}).items():
            os.environ[k] = str(v)
        if c.get("gather"):
            flags |= _lib.MAT_FORCE_GATHER
        else:
            os.environ["DNM_TILE_BITS"] = str(c["B"])
            os.environ["DNM_LOG_ROWS"] = str(c["R"])
            os.environ["DNM_PLAN_MODE"] = str(c["mode"])
            os.environ["DNM_AMIN"] = str(c["amin"])
            os.environ["DNM_GBITS"] = str(c.get("g", 5))
            os.environ["DNM_CACHE_POLICY"] = str(c.get("cp", 0))
            if c.get("glds", 0):
                flags |= _lib.MAT_USE_GLDS
        H = operator_for(c)
        assert int(c.get("env", {}).get("DNM_SWZ", config.vec_swizzle)) == config.vec_swizzle, "one layout per process: export DNM_SWZ"
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c(),
                                flags=flags)
        if os.environ.get("PROBE_DESCRIBE"):
            print(mat.describe(), flush=True)
        nl = C.c_int()
        _lib.lib().dnm_mat_plan_launches(mat.handle, C.byref(nl))
        for _ in range(2):
            mat.mult(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            mat.mult(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        # cheap cross-check between configurations: <x|y> must agree
        d = y.dot(x)
        if ref is None or "bonds" in c:
            ref = d
        print("L=%d %-44s launches=%d  %8.3f ms  %7.2f Gamp/s  %6.1f GB/s(32B)  frac=%.3f  dchk=%.2e" % (
            L, json.dumps(c, separators=(',', ':')), nl.value, ms, dim / ms / 1e6, 32.0 * dim / ms / 1e6,
            32.0 * dim / ms / 1e6 / 8000.0, abs(d - ref)), flush=True)
        mat.destroy()


if __name__ == "__main__":
    main()
