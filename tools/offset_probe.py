"""Does the relative placement of x and y in HBM matter for the L=30 multiply?  y is placed at a byte offset inside a
larger allocation (channel / bank interleaving of reads against writes).   python tools/offset_probe.py [L=30]"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynamite_amd import models, backend, msc_tools
from dynamite_amd.config import config
from dynamite_amd.subspaces import Full

L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
config._initialize()
H = models.mbl(L)
H.reduce_msc()
masks, offs = msc_tools.get_mask_offsets(H.msc)
sub = Full(L=L)
mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c())
n = 1 << L
pad = 1 << 22                      # amplitudes of slack (64 MiB)
xbuf = torch.empty(n + pad, dtype=torch.complex128, device=config.device)
ybuf = torch.empty(n + pad, dtype=torch.complex128, device=config.device)
print("base addresses: x %#x  y %#x" % (xbuf.data_ptr(), ybuf.data_ptr()))
x = backend.Vec(n, array=xbuf[:n], swz=sub.vec_swizzle)
x.set_random(0)
for off_bytes in (0, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256, 3 << 20, 33 << 20, 0):
    o = off_bytes // 16
    y = backend.Vec(n, array=ybuf[o:o + n], swz=sub.vec_swizzle)
    for _ in range(3):
        mat.mult(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        mat.mult(x, y)
    e1.record()
    torch.cuda.synchronize()
    print("y offset %9d B: %.3f ms" % (off_bytes, e0.elapsed_time(e1) / 20), flush=True)
