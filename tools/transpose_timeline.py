#!/usr/bin/env python
"""Per-phase timeline of the transposed exchange on ONE GPU: P ranks share the device (gloo, blocks staged through
the host -- RCCL refuses two ranks on one device), rank 0 records when every phase of `ShellMat._mult_transposed`
is done.  The transfers here are host copies, so their durations say nothing about xGMI; what the timeline shows is
the ORDER of the phases, the device time of each compute piece at the chosen size, and how much of the `y += w`
sweep is left after the last returned batch.  usage: transpose_timeline.py [L P]   (default 27 4: 2^25 amplitudes per rank)"""
import os
os.environ.setdefault("DNM_EXPERIMENTAL", "1")   # tools drive experiment knobs
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, L):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynamite_amd import models
    from dynamite_amd.config import config
    from dynamite_amd.states import State
    config.L = L
    config._initialize()
    H = models.mbl(L)
    x = State(L=L, state='random', seed=1)
    y = State(L=L)
    mat = H.get_mat()
    assert mat.exchange_summary()["scheme"] == "transpose"
    for _ in range(2):
        mat.mult(x.vec, y.vec)
    mat.trace = []
    mat.mult(x.vec, y.vec)
    if rank == 0:
        print("transposed exchange, L=%d on %d ranks sharing one GPU (host-staged transport): 2^%d amplitudes per rank"
              % (L, world, L - (world.bit_length() - 1)))
        print("%-62s %10s %10s" % ("phase (device idle after it)", "at ms", "took ms"))
        prev = 0.0
        for name, t in mat.trace:
            print("%-62s %10.2f %10.2f" % (name, t * 1e3, (t - prev) * 1e3))
            prev = t
    dist.barrier()
    dist.destroy_process_group()


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(P, port, L), nprocs=P, join=True)


if __name__ == "__main__":
    main()
