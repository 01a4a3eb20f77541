"""One rank's compute share of the transposed exchange (backend.ShellMat._mult_transposed) at full per-rank size on
ONE GPU: rank R of P for L = n_loc + log2(P), no communication -- the rank-local operator in the state's layout,
the operator of the redistributed layout, the sum.  What is left of a multi-GPU step is the two all-to-alls.

    python tools/transpose_probe.py [n_loc=30] [P=8] [rank=5]
"""
import ctypes as C
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dynamite_amd import _lib, backend, models, msc_tools
from dynamite_amd.subspaces import Full

nloc_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 30
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
p = P.bit_length() - 1
L = nloc_bits + p
H = models.mbl(L)
H.reduce_msc()
masks, offs = msc_tools.get_mask_offsets(H.msc)
arrs = (masks, offs, H.msc['signs'], H.msc['coeffs'])
sub = Full(L=L)
S = min(int(os.environ.get("DNM_SWZ", "16")), (L - 2 * p - 1 - 2 + 4) // 2)      # DNM_SWZ picks the shift to try
sc = sub._c()
sc.vec_swizzle = S
split = backend.transpose_split(*arrs, L, P, S)
assert split is not None
lo, hi, f = split
Lb = _lib.lib()
n = 1 << nloc_bits
x = backend.Vec(n, swz=S); x.start = R * n; x.set_random(1)
y = backend.Vec(n, swz=S)
xb = backend.Vec(n, swz=S); xb.start = 0; xb.set_random(2)
wb = backend.Vec(n, swz=S)
hl = backend.create_mat(*lo, sc, sc, False, 0, R, P)
hh_old = backend.create_mat(*hi, sc, sc, False, 0, R, P)         # the planner's own tile (round 2)
# tiles [0, a) + [f, n): ranges of workgroups are contiguous parts of the pieces (ShellMat.set_transposed)
hh = backend.create_mat(*hi, sc, sc, False, (12 - (nloc_bits - f)) << _lib.MAT_AMIN_SHIFT, R, P)
full = backend.create_mat(*arrs, sc, sc, False, 0, R, P)
for name, h in (("whole operator (partner scheme)", full), ("layout A (no top spin flipped)", hl), ("layout B (top spins)", hh)):
    buf = C.create_string_buffer(8192)
    _lib.check(Lb.dnm_mat_plan_describe(h, buf, len(buf)))
    print(name, ":", buf.value.decode().strip().replace("\n", " | "))
snd, rcv = backend.exchange_plan(full)
print("partner scheme, rank %d of %d: receives %s GiB from partners %s (largest from one partner %.1f GiB)" % (
    R, P, sum(16 * c for _, _, c in rcv) / 2**30, sorted({q for q, _, _ in rcv}),
    max([sum(16 * c for q2, _, c in rcv if q2 == q) for q in {q for q, _, _ in rcv}] or [0]) / 2**30))
pieces, own, cnt = backend.transpose_pieces(nloc_bits, p, f, R)
print("transposed scheme: %d pieces of %.2f GiB to/from each of %d peers, twice per multiply: %.1f GiB per link" % (
    len(pieces) // (P - 1), 16 * cnt / 2**30, P - 1, 2 * 16 * cnt * (len(pieces) // (P - 1)) / 2**30))


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


t_lo = timed(lambda: _lib.check(Lb.dnm_mat_mult_local(hl, x.ptr, y.ptr, None)))
t_hi = timed(lambda: _lib.check(Lb.dnm_mat_mult_local(hh, xb.ptr, wb.ptr, None)))
t_hi_old = timed(lambda: _lib.check(Lb.dnm_mat_mult_local(hh_old, xb.ptr, wb.ptr, None)))
t_hi_parts = [timed(lambda s=s: _lib.check(Lb.dnm_mat_mult_local_part(hh, xb.ptr, wb.ptr, s, 4, None))) for s in range(4)]
print("layout-B pass: %.2f ms with tiles [0,a)+[f,n) (in 4 ranges: %s ms), %.2f ms with the planner's own tile"
      % (t_hi, " / ".join("%.2f" % t for t in t_hi_parts), t_hi_old))
t_add = timed(lambda: _lib.check(Lb.dnm_vec_axpby(y.ptr, wb.ptr, n, 1.0, 0.0, 1.0, 0.0, None)))
t_cp = timed(lambda: [_lib.check(Lb.dnm_vec_copy(C.c_void_p(x.array[o:o + cnt].data_ptr()),
                                                 C.c_void_p(xb.array[o:o + cnt].data_ptr()), cnt, None)) for o in own])
t_full = timed(lambda: _lib.check(Lb.dnm_mat_mult_local(full, x.ptr, y.ptr, None)))
print("n_loc=%d P=%d rank=%d swizzle=%d: layout-A passes %.2f ms, layout-B pass %.2f ms, sum %.2f ms, own pieces %.2f ms; "
      "rank-local part of the partner scheme %.2f ms" % (nloc_bits, P, R, S, t_lo, t_hi, t_add, t_cp, t_full))
link = float(os.environ.get("XGMI_GBS", "64"))
a2a = 16 * cnt * (len(pieces) // (P - 1)) / (link * 1e9) * 1e3
print("predicted step at %.0f GB/s per link and direction, whole pieces: max(all-to-all %.1f, A %.1f) + B %.1f + all-to-all %.1f + sum %.1f = %.1f ms"
      % (link, a2a, t_lo, t_hi, a2a, t_add, max(a2a, t_lo) + t_hi + a2a + t_add))
print("pipelined in 4 parts: 2 x %.1f + last range of B %.1f (if not hidden behind the first returns) + last addition %.1f = %.1f ms"
      % (a2a, t_hi_parts[-1], t_add / 4, 2 * a2a + t_hi_parts[-1] + t_add / 4))
worst = max([sum(16 * c for q2, _, c in rcv if q2 == q) for q in {q for q, _, _ in rcv}] or [0])
print("partner scheme, same link rate: busiest link %.1f ms + its pass" % (worst / (link * 1e9) * 1e3))
