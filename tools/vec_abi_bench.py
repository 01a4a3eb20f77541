#!/usr/bin/env python
"""Times the vector sweeps of the C ABI at 2^L amplitudes: vec_abi_bench.py [L]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamite_amd import _lib  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n = 1 << L
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.zeros(n, dtype=torch.complex128, device="cuda")
    y = torch.zeros(n, dtype=torch.complex128, device="cuda")
    V = torch.zeros(4 * n, dtype=torch.complex128, device="cuda")
    xp, yp, Vp = (C.c_void_p(t.data_ptr()) for t in (x, y, V))
    out = (C.c_double * 16)()
    coef = (C.c_double * 8)(*([0.5, 0.0] * 4))

    def timed(name, f, bytes_per_amp, reps=5):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-34s %8.3f ms  %7.1f GB/s (%d B/amp)" % (name, ms, bytes_per_amp * n / 1e6 / ms, bytes_per_amp), flush=True)

    timed("dnm_vec_set", lambda: _lib.check(lib.dnm_vec_set(xp, n, 1.0, 0.0, st)), 16)
    timed("dnm_vec_copy (copy kernel)", lambda: _lib.check(lib.dnm_vec_copy(xp, yp, n, st)), 32)
    timed("torch copy_ (device memcpy)", lambda: y.copy_(x), 32)
    timed("dnm_vec_scale", lambda: _lib.check(lib.dnm_vec_scale(xp, n, 0.5, 0.25, st)), 32)
    timed("dnm_vec_axpby", lambda: _lib.check(lib.dnm_vec_axpby(yp, xp, n, 0.5, 0.0, 0.25, 0.0, st)), 48)
    timed("dnm_vec_dot", lambda: _lib.check(lib.dnm_vec_dot(xp, yp, n, out, st)), 32)
    timed("dnm_vec_norm2", lambda: _lib.check(lib.dnm_vec_norm2(xp, n, out, st)), 16)
    timed("dnm_vec_mdot (4 vectors)", lambda: _lib.check(lib.dnm_vec_mdot(Vp, n, 4, xp, n, out, st)), 80)
    timed("dnm_vec_maxpy (4 vectors)", lambda: _lib.check(lib.dnm_vec_maxpy(yp, Vp, n, 4, n, coef, st)), 96)


if __name__ == "__main__":
    main()
