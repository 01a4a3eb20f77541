#!/usr/bin/env python
"""Where a workgroup of tile_pass_kernel spends its life: phase stamps (wall clock, 100 MHz) of thread 0 of every
workgroup, from a diagnostic build (DNM_HIPCC_EXTRA=-DDNM_PHASE_TIMING python -m dynamite_amd.build --force).
usage: python tools/phase_times.py L"""
import ctypes as C
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from dynamite_amd import models, backend, msc_tools, _lib  # noqa: E402
from dynamite_amd.config import config  # noqa: E402
from dynamite_amd.subspaces import Full  # noqa: E402

NAMES = ["tile loads landed, written to LDS", "gather records", "barrier (+ diagonal part 1)", "diagonal part 2",
         "LDS records", "late y load + add", "stores acknowledged"]


def main():
    L = int(sys.argv[1])
    config._initialize()
    sub = Full(L=L)
    dim = 1 << L
    x, y = backend.Vec(dim, swz=sub.vec_swizzle), backend.Vec(dim, swz=sub.vec_swizzle)
    x.set_random(0)
    H = models.BY_NAME["mbl"](L)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c())
    print(mat.describe())
    nblk = dim >> 12
    buf = torch.zeros(2 * nblk * 8, dtype=torch.int64, device="cuda")
    lib = _lib.lib()
    lib.dnm_debug_phase_buffer.argtypes = [C.c_void_p]
    for _ in range(3):
        mat.mult(x, y)
    torch.cuda.synchronize()
    assert lib.dnm_debug_phase_buffer(C.c_void_p(buf.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mat.mult(x, y)
    e1.record()
    torch.cuda.synchronize()
    print("multiply with stamps: %.3f ms" % e0.elapsed_time(e1))
    t = buf.cpu().numpy().reshape(2, nblk, 8).astype("float64") * 10.0      # ns (100 MHz)
    for p in range(2):
        d = t[p, :, 1:] - t[p, :, :-1]
        life = t[p, :, 7] - t[p, :, 0]
        span = (t[p, :, 7].max() - t[p, :, 0].min()) / 1e6
        print("pass %d: %d workgroups, kernel span %.3f ms, workgroup life mean %.2f us (median %.2f), resident on average %.0f"
              % (p, nblk, span, life.mean() / 1e3, float(sorted(life)[nblk // 2]) / 1e3, life.sum() / 1e6 / span))
        for i, nme in enumerate(NAMES):
            print("   %-36s mean %6.2f us   median %6.2f us" % (nme, d[:, i].mean() / 1e3, float(sorted(d[:, i])[nblk // 2]) / 1e3))
    mat.destroy()


if __name__ == "__main__":
    main()
