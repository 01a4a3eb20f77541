#!/usr/bin/env python
"""Does an RCCL transfer run AT THE SAME TIME as a kernel on another stream on this system?  One process, world size 1
(nccl backend), a send-to-self of 4 GiB on RCCL's stream against (a) a compute-bound kernel (fp64 matrix product: no HBM
contention) and (b) a memory-bound one (a copy of 8 GiB) on the current stream: each alone, then together.
together ~ max(alone): concurrent; together ~ sum: serialised (or, for (b), both bound by the same HBM)."""
import datetime
import os
import socket
import time

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                  HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
import sys
opts = None
if "--high-priority" in sys.argv:          # torch's RCCL stream with high priority: a hardware queue of its own class
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=300),
                        pg_options=opts)
dev = torch.device("cuda", 0)
n = 1 << 28                                   # 4 GiB of complex128
src = torch.ones(n, dtype=torch.complex128, device=dev)
dst = torch.empty_like(src)
a = torch.randn(6144, 6144, dtype=torch.float64, device=dev)
b = torch.randn(6144, 6144, dtype=torch.float64, device=dev)
c = torch.empty_like(a)
big = torch.empty(1 << 29, dtype=torch.complex128, device=dev)
big2 = torch.empty_like(big)


def xfer():
    return dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)])


def timed(fn, n=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def alone_x():
    for r in xfer():
        r.wait()


def mm():
    for _ in range(3):
        torch.mm(a, b, out=c)


def cp():
    big2.copy_(big)


def both(k):
    def f():
        reqs = xfer()
        k()
        for r in reqs:
            r.wait()
    return f


tx, tm, tc = timed(alone_x), timed(mm), timed(cp)
print("RCCL send-to-self of 4 GiB alone: %.2f ms; fp64 matrix products alone: %.2f ms; 8 GiB copy alone: %.2f ms" % (tx, tm, tc))
print("transfer + matrix products together: %.2f ms   (max %.2f, sum %.2f)" % (timed(both(mm)), max(tx, tm), tx + tm))
print("transfer + copy together:            %.2f ms   (max %.2f, sum %.2f)" % (timed(both(cp)), max(tx, tc), tx + tc))
dist.destroy_process_group()
