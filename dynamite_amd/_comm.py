"""
Transport for the few collectives the partitioned path needs, over torch.distributed.

With the "nccl" backend (RCCL over xGMI) device tensors go straight to the collective.  The "gloo" backend
moves host memory only (apart from all_reduce / broadcast), so device tensors are staged through host copies
there: that is how several ranks can share one GPU -- RCCL refuses two ranks on one device -- which is what
the 2- and 3-process end-to-end tests in tests/test_gpu_distributed.py do on a single-GPU box.  Only the
transport differs; every kernel runs on the GPU either way.
"""
import torch
import torch.distributed as dist


def nccl_options():
    """Options for ``dist.init_process_group("nccl", pg_options=...)``: RCCL's stream with HIGH priority.  On MI355X / ROCm 7
    streams of one priority share the process's few hardware queues, and a transfer on one stream and a kernel on another
    ran one after the other whenever the two landed on the same queue -- no overlap of exchange and compute at all
    (tools/probes/rccl_concurrency_probe.py: 25.1 ms together = 3.6 + 20.7; 21.8 with this option).  The library's own
    exchange stream (csrc/comm.cpp) is created with the highest priority for the same reason."""
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    return opts


def _staged(t):
    return t.is_cuda and dist.get_backend() == 'gloo'


class _Done:
    def wait(self):
        pass


class _StagedSend:
    """A send from a host copy, which has to outlive the request."""

    def __init__(self, req, host):
        self.req, self.host = req, host

    def wait(self):
        self.req.wait()
        self.host = None


class _StagedRecv:
    """A receive into a host buffer that lands in its device tensor when waited for."""

    def __init__(self, dev, host):
        self.dev, self.host, self.req = dev, host, None

    def wait(self):
        self.req.wait()
        self.dev.copy_(self.host)


def batch_p2p(sends, recvs):
    """Post every (tensor, peer) send and receive as one batch; returns objects with ``wait()``.
    Between a pair of ranks messages match in the order they are listed on both sides."""
    if not sends and not recvs:
        return []
    staged = any(_staged(t) for t, _ in list(sends) + list(recvs))
    if not staged:
        ops = [dist.P2POp(dist.isend, t, p) for t, p in sends]
        ops += [dist.P2POp(dist.irecv, t, p) for t, p in recvs]
        return dist.batch_isend_irecv(ops)
    keep = [t.cpu() for t, _ in sends]                 # host copies stay alive until the sends complete
    landing = [_StagedRecv(t, torch.empty(t.shape, dtype=t.dtype)) for t, _ in recvs]
    ops = [dist.P2POp(dist.isend, h, p) for h, (_, p) in zip(keep, sends)]
    ops += [dist.P2POp(dist.irecv, r.host, p) for r, (_, p) in zip(landing, recvs)]
    reqs = dist.batch_isend_irecv(ops)
    out = []
    if len(reqs) == len(ops):
        out = [_StagedSend(rq, h) for rq, h in zip(reqs[:len(keep)], keep)]
        for r, rq in zip(landing, reqs[len(keep):]):
            r.req = rq
            out.append(r)
    else:       # a backend that returns one request for the whole batch
        class _All:
            def __init__(self, rqs, lands):
                self.rqs, self.lands = rqs, lands

            def wait(self):
                for q in self.rqs:
                    q.wait()
                for ld in self.lands:
                    ld.dev.copy_(ld.host)
        out = [_All(reqs, landing)]
        out[0].keep = keep
    return out


def all_gather(tensor):
    """List of every rank's ``tensor`` (same shape everywhere), on the tensor's device."""
    ws = dist.get_world_size()
    if _staged(tensor):
        h = tensor.cpu()
        parts = [torch.empty_like(h) for _ in range(ws)]
        dist.all_gather(parts, h)
        return [p.to(tensor.device) for p in parts]
    parts = [torch.empty_like(tensor) for _ in range(ws)]
    dist.all_gather(parts, tensor)
    return parts


def gather_varied(tensor, sizes, dst=0):
    """Rank ``dst`` gets the list of every rank's 1-D tensor (lengths ``sizes``); None elsewhere."""
    me = dist.get_rank()
    staged = _staged(tensor)
    src = tensor.cpu() if staged else tensor
    if me == dst:
        parts = [torch.empty(n, dtype=src.dtype, device=src.device) for n in sizes]
        recvs = [(parts[q], q) for q in range(len(sizes)) if q != dst]
        for r in batch_p2p([], recvs):
            r.wait()
        parts[dst] = src
        return [p.to(tensor.device) for p in parts] if staged else parts
    for r in batch_p2p([(src, dst)], []):
        r.wait()
    return None


def reduce_sum(tensor, dst=0):
    """Sum over the ranks, in place on rank ``dst`` (other ranks' tensors are left unspecified)."""
    flat = torch.view_as_real(tensor) if tensor.is_complex() else tensor     # shares the memory
    if _staged(tensor):
        dist.all_reduce(flat)            # gloo reduces device tensors only through all_reduce
    else:
        dist.reduce(flat, dst=dst)


def barrier():
    dist.barrier()
