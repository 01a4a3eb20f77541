"""
Python face of the native backend: the objects dynamite gets from petsc4py /
its Cython layer on this path.

* ``build_mat`` / ``precompute_diagonal`` mirror ``bpetsc.build_mat`` and
  ``bpetsc.precompute_diagonal`` (reference
  ``src/dynamite/_backend/bpetsc.pyx:78-147``) and return a ``ShellMat`` with
  the ``petsc4py.Mat`` methods the path uses: ``mult``, ``norm``, ``destroy``,
  ``getSize``.
* ``Vec`` carries the ``petsc4py.Vec`` methods ``states.py`` calls
  (``states.py:102-123, 703-797``): a contiguous complex128 block per rank,
  row-block partitioned (PETSc's default layout).

Device memory, streams and the rank exchange come from torch (plumbing); all
arithmetic is done by the HIP kernels behind the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from .config import config, knob


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


_NATIVE_COMM = None


def native_comm():
    """The library's own RCCL communicator over the ranks of the torch.distributed process group (dnm_comm_create,
    csrc/comm.cpp), made on first use: rank 0 draws the 128-byte id, the process group's store hands it round.
    Collective.  One rank: a communicator of one (dnm_comm_loopback can make it stand for a rank of many: tests)."""
    global _NATIVE_COMM
    if _NATIVE_COMM is None:
        config._initialize()
        ident = (C.c_char * 128)()
        d = _dist()
        if d is None or config.rank == 0:
            _lib.check(_lib.lib().dnm_comm_unique_id(ident))
        if d is not None:
            box = [bytes(ident.raw)]
            d.broadcast_object_list(box, src=0)
            ident = (C.c_char * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        _lib.check(_lib.lib().dnm_comm_create(ident, config.rank, config.world_size, C.byref(h)))
        _NATIVE_COMM = h
    return _NATIVE_COMM


def release_native_comm():
    """Destroy the library's communicator (ncclCommDestroy, the exchange stream): collective in spirit -- every rank
    calls it, after the last operator that used it has been destroyed and before the process group goes."""
    global _NATIVE_COMM
    if _NATIVE_COMM is not None:
        h, _NATIVE_COMM = _NATIVE_COMM, None
        _lib.check(_lib.lib().dnm_comm_destroy(h))


def native_transport():
    """Whether partitioned multiplies (and the solvers' hooks) go through the library's own communicator:
    config.native_comm, by default whenever the ranks talk over RCCL."""
    d = _dist()
    if d is None:
        return False
    if config.native_comm is None:
        return d.get_backend() == 'nccl'
    return bool(config.native_comm)


def split_ownership(size, world, rank):
    """PetscSplitOwnership: size // world entries each, the first size % world ranks one more.
    Returns (start, local_size)."""
    q, rem = divmod(int(size), world)
    return rank * q + min(rank, rem), q + (1 if rank < rem else 0)


def layout_partition(sub_c, nranks, rank):
    """(istart, ilen, nstart, nlen) of rank's part of a SpinConserve vector in the internal layout: positions
    [istart, istart + ilen) of the layout = indices [nstart, nstart + nlen) of the reference order."""
    v = [C.c_int64() for _ in range(4)]
    _lib.check(_lib.lib().dnm_vec_layout_partition(C.byref(sub_c), int(nranks), int(rank), *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def redistribute(local, src_parts, dst_parts, me):
    """Move a vector between two contiguous partitions of the same index order: ``local`` is this rank's block under
    ``src_parts`` (a list of (start, n) per rank), the result its block under ``dst_parts``.  Collective; overlaps
    of a rank's old and new block are copied in place, the rest travels as one batch of sends and receives (between a
    pair of ranks at most one message each way, so in-order matching is trivially right)."""
    import torch
    from . import _comm
    s0, sn = src_parts[me]
    d0, dn = dst_parts[me]
    if local.numel() != sn:
        raise ValueError('redistribute: the local block has %d elements, its partition says %d' % (local.numel(), sn))
    out = torch.empty(dn, dtype=local.dtype, device=local.device)
    sends, recvs = [], []
    for q in range(len(src_parts)):
        lo, hi = max(s0, dst_parts[q][0]), min(s0 + sn, dst_parts[q][0] + dst_parts[q][1])      # mine -> q's new block
        if lo < hi:
            if q == me:
                out[lo - d0:hi - d0] = local[lo - s0:hi - s0]
            else:
                sends.append((local[lo - s0:hi - s0], q))
        lo, hi = max(d0, src_parts[q][0]), min(d0 + dn, src_parts[q][0] + src_parts[q][1])      # q's old block -> mine
        if lo < hi and q != me:
            recvs.append((out[lo - d0:hi - d0], q))
    for r in _comm.batch_p2p(sends, recvs):
        r.wait()
    return out


def layout_blocks(sub_c):
    """(T, ibase): the T blocks of a SpinConserve internal layout in the order they lie in memory and their first
    positions (ibase has one entry more: the layout's size) -- dnm_vec_layout_blocks, host tables."""
    n = C.c_int64()
    _lib.check(_lib.lib().dnm_vec_layout_blocks(C.byref(sub_c), 0, None, None, C.byref(n)))
    T = np.zeros(n.value, dtype=np.int64)
    ib = np.zeros(n.value + 1, dtype=np.int64)
    _lib.check(_lib.lib().dnm_vec_layout_blocks(C.byref(sub_c), n.value, _lib.p64(T), _lib.p64(ib), C.byref(n)))
    return T, ib


def block_moves(d_from, d_to, nranks):
    """What moving a partitioned vector between two layouts that differ in BLOCK ORDER only (vec_swizzle bits 16-19) takes:
    per block in ascending T, int64 arrays (src rank, offset inside the source rank's share, dst rank, offset inside the
    destination's share, positions).  Blocks move whole, padding included."""
    Tf, ibf = layout_blocks(d_from)
    Tt, ibt = layout_blocks(d_to)
    of, ot = np.argsort(Tf, kind='stable'), np.argsort(Tt, kind='stable')
    lf, lt = np.diff(ibf)[of], np.diff(ibt)[ot]
    if (int(d_from.vec_swizzle) ^ int(d_to.vec_swizzle)) & 0xffff or not np.array_equal(Tf[of], Tt[ot]) \
            or not np.array_equal(lf, lt):
        raise ValueError('block_moves: the two layouts differ in more than the order of their blocks')
    sf = np.array([layout_partition(d_from, nranks, q)[0] for q in range(nranks)], dtype=np.int64)
    st = np.array([layout_partition(d_to, nranks, q)[0] for q in range(nranks)], dtype=np.int64)
    pf, pt = ibf[:-1][of], ibt[:-1][ot]
    # (a rank without blocks starts where the next one does: the LAST rank starting at or before a block owns it)
    src = np.searchsorted(sf, pf, side='right') - 1
    dst = np.searchsorted(st, pt, side='right') - 1
    return src, pf - sf[src], dst, pt - st[dst], lf


def reorder_blocks(local, d_from, d_to, per_position=2):
    """This rank's share of a vector in layout ``d_to`` from its share ``local`` in layout ``d_from`` -- the same
    SpinConserve internal layout up to the order of its T blocks (the reference-compatible one that States live in, and
    the one made for partitions that solvers iterate in: Operator.get_solver_mat).  Collective.  ``per_position``:
    doubles per position of the layout (2: complex128 vectors; 1: real vectors stored two positions to an element).
    One message each way between any two ranks per round of a ring schedule (round s: to rank me + s, from rank me - s);
    a round stages what it sends and receives, at most one share."""
    import torch
    from . import _comm
    ws, me = config.world_size, config.rank
    src, soff, dst, doff, ln = block_moves(d_from, d_to, ws)
    u = int(per_position)
    f = torch.view_as_real(local).reshape(-1)
    n_from, n_to = layout_partition(d_from, ws, me)[1], layout_partition(d_to, ws, me)[1]
    if f.numel() != u * n_from:
        raise ValueError('reorder_blocks: the local share holds %d doubles, its layout says %d' % (f.numel(), u * n_from))
    out = torch.empty(u * n_to, dtype=f.dtype, device=f.device)
    mine = np.nonzero((src == me) & (dst == me))[0]
    for j in mine:
        out[u * doff[j]:u * (doff[j] + ln[j])] = f[u * soff[j]:u * (soff[j] + ln[j])]
    for s in range(1, ws):
        to, frm = (me + s) % ws, (me - s) % ws
        js = np.nonzero((src == me) & (dst == to))[0]
        jr = np.nonzero((src == frm) & (dst == me))[0]
        sends, recvs, rbuf = [], [], None
        if js.size:
            sends.append((torch.cat([f[u * soff[j]:u * (soff[j] + ln[j])] for j in js]), to))
        if jr.size:
            rbuf = torch.empty(u * int(ln[jr].sum()), dtype=f.dtype, device=f.device)
            recvs.append((rbuf, frm))
        for r in _comm.batch_p2p(sends, recvs):
            r.wait()
        at = 0
        for j in jr:
            out[u * doff[j]:u * (doff[j] + ln[j])] = rbuf[at:at + u * ln[j]]
            at += u * int(ln[j])
    return torch.view_as_complex(out.reshape(-1, 2))


def window_exchange_ops(owned, windows, me, needs=None):
    """Who sends what to whom so that every rank holds the columns of its window.
    owned[q] = (start, n) of rank q's block; windows[q] = inclusive (cmin, cmax) rank q reads; needs[q] (optional)
    = the ascending, disjoint [lo, hi) ranges inside that window rank q really reads (``needed_ranges``) --
    without it the whole window travels.
    Returns (recvs, sends): recvs = [(src, lo, hi)] global index ranges [lo, hi) to receive from
    src, sends = [(dst, lo, hi)] ranges of MY block to send; between a pair of ranks both lists ascend."""
    def wanted(q):
        return needs[q] if needs is not None else [(windows[q][0], windows[q][1] + 1)]
    recvs, sends = [], []
    my0, myn = owned[me]
    for q, (q0, qn) in enumerate(owned):
        if q == me:
            continue
        for wlo, whi in wanted(me):
            lo, hi = max(wlo, q0), min(whi, q0 + qn)
            if lo < hi:
                recvs.append((q, lo, hi))
        for qlo, qhi in wanted(q):
            lo, hi = max(qlo, my0), min(qhi, my0 + myn)
            if lo < hi:
                sends.append((q, lo, hi))
    return recvs, sends


def complement_ranges(ranges, n):
    """The parts of [0, n) that the ascending, disjoint ``ranges`` [(a, b), ...] leave out."""
    out, at = [], 0
    for a, b in ranges:
        if not (at <= a < b <= n):
            raise ValueError('ranges must be ascending, disjoint and inside [0, %d): %r' % (n, ranges))
        if a > at:
            out.append((at, a))
        at = b
    if at < n:
        out.append((at, n))
    return out


def needed_ranges(chunk_map, shift, window):
    """[lo, hi) column ranges covering the marked chunks of ``dnm_mat_column_chunks`` (clipped to the window)."""
    wlo, whi = window[0], window[1] + 1
    first = wlo >> shift
    marked = np.flatnonzero(np.asarray(chunk_map, dtype=np.uint8))
    if marked.size == 0:
        return []
    cuts = np.flatnonzero(np.diff(marked) > 1)
    starts = np.concatenate(([marked[0]], marked[cuts + 1]))
    ends = np.concatenate((marked[cuts], [marked[-1]]))
    return [(max(wlo, int(first + a) << shift), min(whi, int(first + b + 1) << shift)) for a, b in zip(starts, ends)]


def post_window_exchange(x_local, owned, windows, me, window_buf=None, needs=None):
    """Post the transfers that assemble this rank's column window from the owners' blocks (torch.distributed
    send/recv: RCCL over xGMI on GPUs, gloo in CPU tests) and copy the rank's own part in.  With ``needs`` only
    the ranges a rank reads are moved; the rest of its window buffer keeps whatever it held (zeros from the
    allocation).  Returns (window buffer, requests to wait for)."""
    import torch
    from . import _comm
    wlo, whi = windows[me][0], windows[me][1] + 1
    if window_buf is None or window_buf.numel() != whi - wlo:
        window_buf = torch.zeros(whi - wlo, dtype=x_local.dtype, device=x_local.device)
    my0, myn = owned[me]
    recvs, sends = window_exchange_ops(owned, windows, me, needs)
    reqs = _comm.batch_p2p([(x_local[lo - my0:hi - my0], q) for q, lo, hi in sends],
                           [(window_buf[lo - wlo:hi - wlo], q) for q, lo, hi in recvs])
    a, b = max(wlo, my0), min(whi, my0 + myn)
    dst, src = window_buf[a - wlo:b - wlo], x_local[a - my0:b - my0]
    if src.is_cuda and src.dtype == torch.complex128 and src.numel():
        # (the library's copy kernel streams at 6.5 TB/s, a device-to-device memcpy at 4.7: profiles/r03_vec_abi.txt)
        _lib.check(_lib.lib().dnm_vec_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel(), _stream()))
    else:
        dst.copy_(src)
    return window_buf, reqs


def exchange_window(x_local, owned, windows, me, window_buf=None, needs=None):
    """``post_window_exchange`` and the wait for its transfers."""
    window_buf, reqs = post_window_exchange(x_local, owned, windows, me, window_buf, needs)
    for r in reqs:
        r.wait()
    return window_buf


def device_zeros(n, empty=False):
    """complex128 device buffer; if HBM is exhausted the solvers' cached Krylov workspace (tens of GiB
    after a large solve) is handed back and the allocation retried once."""
    import torch
    make = torch.empty if empty else torch.zeros
    try:
        return make(n, dtype=torch.complex128, device=config.device)
    except torch.OutOfMemoryError:
        _lib.check(_lib.lib().dnm_release_workspace())
        torch.cuda.empty_cache()
        return make(n, dtype=torch.complex128, device=config.device)


def exchange_plan(handle):
    """(sends, recvs) of the partitioned multiply: lists of (partner, offset, count),
    offsets/counts in amplitudes of the SENDER's local vector; recvs[i] feeds
    dnm_mat_mult_remote(handle, i, ...)."""
    ns, nr = C.c_int(), C.c_int()
    L = _lib.lib()
    _lib.check(L.dnm_mat_exchange_plan(handle, C.byref(ns), None, C.byref(nr), None))
    sb, rb = (_lib.Xfer * max(1, ns.value))(), (_lib.Xfer * max(1, nr.value))()
    _lib.check(L.dnm_mat_exchange_plan(handle, C.byref(ns), sb, C.byref(nr), rb))
    sends = [(int(sb[i].partner), int(sb[i].offset), int(sb[i].count)) for i in range(ns.value)]
    recvs = [(int(rb[i].partner), int(rb[i].offset), int(rb[i].count)) for i in range(nr.value)]
    return sends, recvs


def post_exchange(x_local, sends, recvs, recv_bufs):
    """Post every send/receive of one multiply as one batch (RCCL send/recv on a
    GPU, gloo on CPU).  Between a pair of ranks the sends of one side are listed in
    the order of the other side's receives, so in-order matching pairs them up."""
    from . import _comm
    return _comm.batch_p2p([(x_local[off:off + cnt], p) for p, off, cnt in sends],
                           [(buf, p) for (p, _, _), buf in zip(recvs, recv_bufs)])


def transpose_split(masks, mask_offsets, signs, coeffs, L, nranks, swizzle=0, shift=0, packed=False):
    """Split a Full-space (``shift`` = 0) or Parity (``shift`` = 1: basis index = configuration >> 1, so index bit
    j is spin j + 1 and ``L`` counts the index bits) operator on P = 2^p ranks for the transposed exchange, or None
    if it does not apply.

    A rank's block holds the n = L - p low spins; the rank number is the p top spins.  A mask that flips a top
    spin couples blocks of different ranks.  Instead of shipping a partner block per such mask
    (bpetsc_template_2.c:787-879 scatters the needed entries), the state is redistributed ONCE so that the top
    spins become local: layout B swaps spins [n, n+p) with the local spins F = [f, f+p), f = n - 1 - p (right
    below the top local spin, which the boundary bond touches).  In layout B every mask that flips a top spin is
    rank-local provided it leaves F alone; it is the same MSC term with the two bit fields swapped.  One
    all-to-all of the state, one local pass, one all-to-all of the result: every xGMI link carries
    2 * 2^n / P amplitudes per multiply instead of up to 2^n on one link.

    ``packed``: the two operators will be built in real-packed form (DNM_MAT_REAL_PACKED: index bit 0 becomes the lane
    of an element, every other bit moves down by one): F must leave bit 0 alone and a piece is 2^(f-1) elements.

    Returns (lo, hi, f): ``lo`` = (masks, offsets, signs, coeffs) of the terms that flip no top spin (layout A),
    ``hi`` = the others with bits permuted (layout B)."""
    p = nranks.bit_length() - 1
    n = L - p
    f = n - 1 - p
    if nranks != 1 << p or p < 1 or f < (1 if packed else 0):
        return None
    if swizzle and f - (1 if packed else 0) < 2 * swizzle - 4:
        return None                    # pieces of 2^f amplitudes must keep their internal order in both layouts
    masks = np.asarray(masks, dtype=np.int64)
    offs = np.asarray(mask_offsets, dtype=np.int64)
    signs = np.asarray(signs, dtype=np.int64)
    coeffs = np.asarray(coeffs, dtype=np.complex128)
    tm = np.repeat(masks, np.diff(offs))
    fs, ns = f + shift, n + shift             # the two fields as spins
    top = (tm >> ns) != 0
    fld = np.int64(nranks - 1)
    if not top.any() or top.all() or ((tm[top] >> fs) & fld).any():
        return None

    def swap(v):
        d = ((v >> fs) ^ (v >> ns)) & fld
        return v ^ (d << fs) ^ (d << ns)

    def csr(m, sg, c):
        order = np.argsort(m, kind='stable')
        m, sg, c = m[order], sg[order], c[order]
        um, first = np.unique(m, return_index=True)
        return um, np.append(first, m.size).astype(np.int64), sg, c

    lo = csr(tm[~top], signs[~top], coeffs[~top])
    hi = csr(swap(tm[top]), swap(signs[top]), coeffs[top])
    return lo, hi, f


def transpose_pieces(n, p, f, rank):
    """The all-to-all between layouts A and B as (peer, offset, count) pieces: the piece at ``offset`` of the
    source vector goes to ``peer`` and the piece received from ``peer`` lands at the same ``offset`` of the
    destination vector (the map is its own inverse).  Pieces for one peer are listed in the same order on both
    sides.  Returns (pieces, own): ``own`` = offsets that stay on this rank."""
    P = 1 << p
    nb = 1 << (n - f - p)
    cnt = 1 << f
    pieces = [(q, (b * P + q) << f, cnt) for q in range(P) if q != rank for b in range(nb)]
    own = [(b * P + rank) << f for b in range(nb)]
    return pieces, own, cnt


def post_transpose(src, dst, pieces):
    """Post one all-to-all between the layouts (see ``transpose_pieces``) as one batch."""
    from . import _comm
    return _comm.batch_p2p([(src[off:off + cnt], q) for q, off, cnt in pieces],
                           [(dst[off:off + cnt], q) for q, off, cnt in pieces])


class ExchangeCheckError(RuntimeError):
    """The first multiply of a transposed-exchange operator disagreed with rows recomputed from the operator's
    definition (``ShellMat.selfcheck``)."""


def site_perm_of(sub_c):
    """The site relabelling of a SpinConserve descriptor (dnm_subspace.site_perm) as a tuple, None for the identity."""
    if sub_c is None or not sub_c.site_perm:
        return None
    perm = tuple(int(sub_c.site_perm[i]) for i in range(int(sub_c.L)))
    return None if perm == tuple(range(int(sub_c.L))) else perm


def with_site_perm(sub_c, perm):
    """A copy of a SpinConserve descriptor whose vectors live in the relabelled layout ``perm`` (spin i -> bit
    perm[i]); the copy keeps the permutation array and the original descriptor (whose tables it points into) alive."""
    d = _lib.Subspace.from_buffer_copy(sub_c)
    arr = np.ascontiguousarray(perm, dtype=np.int8)
    d.site_perm = arr.ctypes.data_as(C.POINTER(C.c_int8))
    d._keepalive = (arr, sub_c)
    return d


def choose_site_perm(masks, L, a, w, fix_top=False):
    """(site_perm as an int8 array, hop counts [Lo, W, T, Lo-W, Lo-T, W-T]) -- dnm_sc_choose_site_perm."""
    masks = np.ascontiguousarray(masks, dtype=np.int64)
    perm = np.zeros(int(L), dtype=np.int8)
    counts = (C.c_int32 * 6)()
    _lib.check(_lib.lib().dnm_sc_choose_site_perm(int(L), int(a), int(w), masks.size, _lib.p64(masks), int(bool(fix_top)),
                                                  perm.ctypes.data_as(C.POINTER(C.c_int8)), counts))
    return perm, [int(c) for c in counts]


class Vec:
    """Distributed complex128 vector: this rank's block lives in ``self.array``
    (a 1-D torch tensor on the rank's GPU).  ``swz``: layout of the block (dnm_subspace.vec_swizzle): 0 = element
    i at position i; 5 <= S <= 24 = XOR-swizzled (Full / Parity states); a | w << 8 = the three-field internal
    layout of a SpinConserve subspace (csrc/sc3.h; ``sub_c`` is then that subspace's descriptor, one rank): rows in
    another order plus zero padding, so ``local_size`` (what the vector kernels sweep) exceeds ``rows``.
    Everything index-wise goes through ``positions`` / ``get_local`` / ``set_local``; the BLAS-1 methods do not
    care."""

    def __init__(self, size, array=None, swz=0, sub_c=None):
        import torch
        config._initialize()
        self.size = int(size)
        self.swz = int(swz)
        self.sub_c = sub_c
        self.perm = site_perm_of(sub_c) if int(swz) >= 256 else None
        self.start, self.local_size = split_ownership(self.size, config.world_size, config.rank)
        self.rows = self.local_size              # elements of the vector this rank holds
        self.istart = self.start                 # where this rank's part starts in the layout's own index space
        self._part = None
        self.half = False
        if self.internal:
            if sub_c is None:
                raise ValueError('a vector in the SpinConserve internal layout needs its subspace descriptor')
            full = C.c_int64()
            _lib.check(_lib.lib().dnm_subspace_dim(C.byref(sub_c), C.byref(full)))
            if 2 * self.size == full.value and config.world_size == 1:
                # an XParity vector on top of the subspace: the representatives are the states whose spin L-1 is up --
                # the blocks of the layout whose top bit is clear, its first half (what rank 0 of 2 would own)
                self.half = True
                self._part = _lib.Partition(0, 2)
                self.istart, self.local_size, self.start, self.rows = layout_partition(sub_c, 2, 0)
                if self.rows != self.size:
                    raise ValueError('the layout does not split into halves for this subspace')
            else:
                # whole blocks of equal top bits per rank: a contiguous range of the layout and of the reference order
                self._part = _lib.Partition(config.rank, config.world_size)
                self.istart, self.local_size, self.start, self.rows = layout_partition(sub_c, config.world_size, config.rank)
        if array is None:
            array = device_zeros(self.local_size)
        elif array.numel() != self.local_size:
            raise ValueError('Vec: the array holds %d elements, this rank\'s part of a vector of %d in layout %d has %d'
                             % (array.numel(), self.size, self.swz, self.local_size))
        self.array = array

    @property
    def internal(self):
        """True for the SpinConserve internal layout (rows reordered and padded)."""
        return self.swz >= 256

    # -- layout ------------------------------------------------------------------
    def positions(self, idx):
        """Positions in ``self.array`` of the local elements ``idx`` (int64 tensor or int)."""
        S = self.swz
        if not S:
            return idx
        if self.internal:
            import torch
            if not torch.is_tensor(idx):
                a = np.ascontiguousarray([int(idx)], dtype=np.int64)
                out = np.empty_like(a)
                _lib.check(_lib.lib().dnm_vec_layout_positions_host(C.byref(self.sub_c), C.byref(self._part), 1,
                                                                    _lib.p64(a), _lib.p64(out)))
                return int(out[0])
            idx = idx.to(device=self.array.device, dtype=torch.int64).contiguous()
            pos = torch.empty_like(idx)
            _lib.check(_lib.lib().dnm_vec_layout_positions(C.byref(self.sub_c), C.byref(self._part), idx.numel(),
                                                           C.c_void_p(idx.data_ptr()), C.c_void_p(pos.data_ptr()),
                                                           _stream()))
            return pos
        return idx ^ (((idx >> S) & ((1 << (S - 4)) - 1)) << 4)

    def get_local(self, lo, hi):
        """Device tensor of the local elements [lo, hi) in index order."""
        import torch
        if not self.swz:
            return self.array[lo:hi]
        return self.array[self.positions(torch.arange(lo, hi, device=self.array.device))]

    def set_local(self, lo, hi, values):
        import torch
        if not self.swz:
            self.array[lo:hi] = values
        else:
            self.array[self.positions(torch.arange(lo, hi, device=self.array.device))] = values

    def local_natural(self):
        """This rank's block in index order (a new device tensor when the layout is not index order)."""
        import torch
        if not self.swz:
            return self.array
        if self.internal:
            out = torch.empty(self.rows, dtype=self.array.dtype, device=self.array.device)
            _lib.check(_lib.lib().dnm_vec_layout_copy(C.byref(self.sub_c), C.byref(self._part),
                                                      C.c_void_p(out.data_ptr()), self.ptr, 0, _stream()))
            return out
        out = torch.empty_like(self.array)
        _lib.check(_lib.lib().dnm_vec_swizzle_copy(C.c_void_p(out.data_ptr()), self.ptr, self.local_size,
                                                   self.swz, _stream()))
        return out

    def set_local_natural(self, t):
        """Fill this rank's block from a device tensor in index order."""
        if self.internal:
            t = t.contiguous()
            if t.numel() != self.rows:
                # (on several ranks a reference-order block -- PetscSplitOwnership -- is not this rank's share of the
                # layout -- whole blocks of equal top bits: the kernel would read past the end of the smaller one)
                raise ValueError('set_local_natural: %d elements for a block of %d rows (vectors of different '
                                 'partitions go through backend.redistribute)' % (t.numel(), self.rows))
            _lib.check(_lib.lib().dnm_vec_layout_copy(C.byref(self.sub_c), C.byref(self._part), self.ptr,
                                                      C.c_void_p(t.data_ptr()), 1, _stream()))
        else:
            self.set_local(0, self.rows, t)

    def _zero_padding(self):
        if self.internal:
            _lib.check(_lib.lib().dnm_vec_layout_zero_padding(C.byref(self.sub_c), C.byref(self._part), self.ptr,
                                                              _stream()))

    # -- petsc4py.Vec-like surface -------------------------------------------
    def getSize(self):
        return self.size

    def getLocalSize(self):
        """Elements of the vector this rank owns (hi - lo of getOwnershipRange, the petsc4py contract); the length of
        ``array`` -- what the vector kernels sweep, padding of the SpinConserve internal layout included -- is
        ``alloc_size``."""
        return self.rows

    @property
    def alloc_size(self):
        return self.local_size

    def getOwnershipRange(self):
        return self.start, self.start + self.rows

    @property
    def ptr(self):
        return C.c_void_p(self.array.data_ptr())

    def set(self, value):
        v = complex(value)
        _lib.check(_lib.lib().dnm_vec_set(self.ptr, self.local_size, v.real, v.imag, _stream()))
        if v != 0:
            self._zero_padding()

    @property
    def layout(self):
        """What two vectors must share to be combined element by element: the layout code and, in the SpinConserve
        internal layout, the site relabelling."""
        return (self.swz, self.perm)

    def _in_my_layout(self, other):
        """``other`` (a vector of the same subspace) as a vector laid out like this one: itself, or a copy made
        through the reference order (a state built for one operator's relabelled layout meeting another's)."""
        if other.layout == self.layout:
            return other
        if other.size != self.size or not (self.internal or other.internal):
            raise ValueError('vectors of different layouts (%r, %r)' % (self.layout, other.layout))
        if other.start != self.start or other.rows != self.rows:
            raise ValueError('vectors of different layouts (%r, %r) and partitions' % (self.layout, other.layout))
        tmp = Vec(self.size, swz=self.swz, sub_c=self.sub_c)
        tmp.set_local_natural(other.local_natural())
        return tmp

    def copy(self, result=None):
        if result is None:
            result = Vec(self.size, swz=self.swz, sub_c=self.sub_c)
        if result.layout != self.layout:
            if not (self.internal and result.internal and result.size == self.size and result.start == self.start
                    and result.rows == self.rows):
                raise ValueError('vectors of different layouts')
            result.set_local_natural(self.local_natural())
            return result
        _lib.check(_lib.lib().dnm_vec_copy(self.ptr, result.ptr, self.local_size, _stream()))
        return result

    def scale(self, alpha):
        a = complex(alpha)
        _lib.check(_lib.lib().dnm_vec_scale(self.ptr, self.local_size, a.real, a.imag, _stream()))

    def axpby(self, alpha, beta, x):
        """self = alpha*x + beta*self (VecAXPBY)."""
        a, b = complex(alpha), complex(beta)
        x = self._in_my_layout(x)
        _lib.check(_lib.lib().dnm_vec_axpby(self.ptr, x.ptr, self.local_size, a.real, a.imag,
                                            b.real, b.imag, _stream()))

    def _reduce(self, vals, op='sum'):
        d = _dist()
        if d is None:
            return vals
        import torch
        t = torch.tensor(vals, dtype=torch.float64, device=self.array.device)
        d.all_reduce(t, op=d.ReduceOp.SUM if op == 'sum' else d.ReduceOp.MAX)
        return t.tolist()

    def dot(self, other):
        """VecDot(self, other) = sum_i self_i * conj(other_i)."""
        out = (C.c_double * 2)()
        other = self._in_my_layout(other)
        _lib.check(_lib.lib().dnm_vec_dot(self.ptr, other.ptr, self.local_size, out, _stream()))
        re, im = self._reduce([out[0], out[1]])
        return complex(re, im)

    def norm(self):
        out = (C.c_double * 2)()
        _lib.check(_lib.lib().dnm_vec_dot(self.ptr, self.ptr, self.local_size, out, _stream()))
        re = self._reduce([out[0]])[0]
        return float(np.sqrt(max(re, 0.0)))

    def normalize(self):
        n = self.norm()
        self.scale(1.0 / n)
        return n

    def shift(self, alpha):
        self.array += complex(alpha)
        self._zero_padding()

    def set_random(self, seed):
        if self.internal:
            _lib.check(_lib.lib().dnm_vec_layout_set_random(C.byref(self.sub_c), C.byref(self._part), self.ptr,
                                                            seed & (2 ** 64 - 1), _stream()))
            return
        _lib.check(_lib.lib().dnm_vec_set_random_swz(self.ptr, self.local_size, seed & (2 ** 64 - 1),
                                                     self.start, self.swz, _stream()))

    _CHUNK = 1 << 24

    def set_local_from_numpy(self, arr):
        import torch
        arr = np.ascontiguousarray(arr, dtype=np.complex128)
        assert arr.size == self.rows
        if not self.swz:
            self.array.copy_(torch.from_numpy(arr))
            return
        if self.internal:
            self.set_local_natural(torch.from_numpy(arr).to(self.array.device))
            return
        for lo in range(0, self.local_size, self._CHUNK):      # chunked: no second full-size device buffer
            hi = min(self.local_size, lo + self._CHUNK)
            self.set_local(lo, hi, torch.from_numpy(arr[lo:hi]).to(self.array.device))

    def local_numpy(self):
        if not self.swz:
            return self.array.cpu().numpy()
        if self.local_size <= self._CHUNK or self.internal:
            return self.local_natural().cpu().numpy()
        out = np.empty(self.local_size, dtype=np.complex128)
        for lo in range(0, self.local_size, self._CHUNK):
            hi = min(self.local_size, lo + self._CHUNK)
            out[lo:hi] = self.get_local(lo, hi).cpu().numpy()
        return out

    def to_numpy(self, to_all=False):
        """Gather to rank 0 (or everywhere): State._to_numpy (states.py:403-447)."""
        d = _dist()
        if d is None:
            return self.local_numpy()
        import torch
        from . import _comm
        ws = config.world_size
        if self.internal:
            sizes = [layout_partition(self.sub_c, ws, q)[3] for q in range(ws)]
        else:
            sizes = [split_ownership(self.size, ws, q)[1] for q in range(ws)]
        if len(set(sizes)) == 1:
            parts = _comm.all_gather(self.local_natural())
        else:   # uneven blocks: pad to the largest
            mx = max(sizes)
            pad = torch.zeros(mx, dtype=self.array.dtype, device=self.array.device)
            pad[:self.rows] = self.local_natural()
            parts = [g[:n] for g, n in zip(_comm.all_gather(pad), sizes)]
        if not to_all and config.rank != 0:
            return None
        return torch.cat(parts).cpu().numpy()

    def destroy(self):
        self.array = None


class RawVec:
    """A device array taken as a vector in a matrix's own layout without the bookkeeping of ``Vec`` -- what the solver
    hooks wrap raw work-vector pointers of a real-arithmetic SpinConserve operator in (its vectors are one double per
    position of the layout, i.e. half as many complex128 elements as ``Vec`` would size them)."""
    internal = True
    perm = None

    def __init__(self, array, swz):
        self.array, self.swz = array, int(swz)
        self.local_size = self.rows = array.numel()

    def positions(self, idx):
        S = self.swz
        if S >= 256:
            raise ValueError('no index-wise access to a raw vector of the SpinConserve layout')
        return idx if not S else idx ^ (((idx >> S) & ((1 << (S - 4)) - 1)) << 4)

    @property
    def ptr(self):
        return C.c_void_p(self.array.data_ptr())


class ShellMat:
    """Matrix-free operator handle (PETSc MatShell + shell_context in the reference)."""

    def __init__(self, handle, left_c, right_c, nranks, rank):
        self._h = handle
        self._keep = (left_c, right_c)
        self.nranks, self.rank = nranks, rank
        M, N, m, n = (C.c_int64() for _ in range(4))
        _lib.check(_lib.lib().dnm_mat_sizes(handle, C.byref(M), C.byref(N), C.byref(m), C.byref(n)))
        self.M, self.N, self.m_local, self.n_local = M.value, N.value, m.value, n.value
        rp = C.c_int()
        _lib.check(_lib.lib().dnm_mat_is_real_packed(handle, C.byref(rp)))
        self.real_packed = bool(rp.value)       # DNM_MAT_REAL_PACKED handle: real vectors, sizes in complex128 elements
        # layouts of y and x the matrix works in: the descriptors' where it supports them (Full / Parity swizzle, the
        # SpinConserve internal layout of a same-subspace pair on one rank), reference order otherwise
        ll, lr = C.c_int(), C.c_int()
        _lib.check(_lib.lib().dnm_mat_layouts(handle, C.byref(ll), C.byref(lr)))
        self.swz_left, self.swz_right = ll.value, lr.value
        # ... and the site relabelling of a SpinConserve pair in the internal layout (dnm_subspace.site_perm)
        self.perm_left = site_perm_of(left_c) if ll.value >= 256 else None
        self.perm_right = site_perm_of(right_c) if lr.value >= 256 else None
        self.sends, self.recvs = exchange_plan(handle)
        self.partners = sorted({r[0] for r in self.recvs})
        self._recv = {}
        r0, ml = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().dnm_mat_ownership(handle, C.byref(r0), C.byref(ml)))
        self.row0 = r0.value
        self._windows = None      # partitioned SpinConserve: every rank's column window
        self._needs = None        # ... and the ranges of it each rank really reads
        self._window_buf = None
        self._tr = None           # transposed exchange (set_transposed): (lo handle, hi handle, pieces, own, cnt)
        self._tr_bufs = None
        self._tr_pipe = False     # layout-B pass and both all-to-alls run sub-piece by sub-piece (_mult_transposed_pipelined)
        self._splits = None       # window multiply in a local and a remote part (dnm_mat_window_split)
        self._row_ranges = None   # ... or by rows: (ranges that read only the rank's own block, the others)
        self._msc = None          # (masks, mask_offsets, signs, coeffs, left subspace dict, right subspace dict): selfcheck
        self._check_pending = False
        # the partitioned multiply as one native call (dnm_mat_mult_partitioned: exchange on the library's own RCCL
        # communicator and stream) instead of the schedules below over torch.distributed -- the default on RCCL
        # transports (native_transport()); the transposed exchange too (set_native_transposed)
        self._native = None
        self._native_tr = False   # transposed exchange split and scheduled inside the library (set_native_transposed)

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError('matrix has been destroyed')
        return self._h

    def getSize(self):
        return self.M, self.N

    def createVecs(self):
        return (Vec(self.N, swz=self.swz_right, sub_c=self._keep[1]), Vec(self.M, swz=self.swz_left, sub_c=self._keep[0]))

    def _mult_converted(self, x, y):
        """A vector whose layout is not the matrix's: a state in the SpinConserve internal layout meets a matrix that
        works in reference order (a projection onto / from another subspace, XParity ...), or a matrix that works in
        a relabelled layout of its own (an operator on a bond graph, ``choose_site_perm``).  Multiply on copies made
        through the reference order."""
        xn, yn = x, y
        P = self.nranks

        def parts(v):
            """(block of the internal layout's partition, block of the reference-order partition) per rank: on
            several ranks the two differ (whole top-bit blocks against PetscSplitOwnership) and the copy is
            redistributed between them."""
            return ([layout_partition(v.sub_c, P, q)[2:4] for q in range(P)],
                    [split_ownership(v.size, P, q) for q in range(P)])

        def mine(v, want_swz, want_perm):
            return v.swz == want_swz and v.perm == want_perm
        if not mine(x, self.swz_right, self.perm_right):
            if not x.internal:
                return False
            arr = x.local_natural()
            if self.swz_right == 0:
                if P > 1:
                    arr = redistribute(arr, *parts(x), self.rank)
                xn = Vec(x.size, array=arr, swz=0)
            elif P == 1:
                xn = Vec(x.size, swz=self.swz_right, sub_c=self._keep[1])
                xn.set_local_natural(arr)
            else:
                return False
        if not mine(y, self.swz_left, self.perm_left):
            if not y.internal:
                return False
            if self.swz_left == 0:
                yn = Vec(y.size, swz=0)
            elif P == 1:
                yn = Vec(y.size, swz=self.swz_left, sub_c=self._keep[0])
            else:
                return False
        self.mult(xn, yn)
        if yn is not y:
            arr = yn.array if yn.swz == 0 else yn.local_natural()
            if P > 1:
                lay, ref = parts(y)
                arr = redistribute(arr, ref, lay, self.rank)
            y.set_local_natural(arr)
        return True

    def describe(self):
        buf = C.create_string_buffer(8192)
        _lib.check(_lib.lib().dnm_mat_plan_describe(self.handle, buf, len(buf)))
        return buf.value.decode()

    def mult(self, x, y):
        """y = A x (MatMult).  Partitioned: the partners' blocks travel over RCCL
        send/recv while the rank-local masks are applied; the off-rank masks
        follow once their block has arrived."""
        L = _lib.lib()
        if x.array.data_ptr() == y.array.data_ptr():
            raise ValueError('x and y must be different vectors')
        if ((x.swz, x.perm) != (self.swz_right, self.perm_right) or (y.swz, y.perm) != (self.swz_left, self.perm_left)) \
                and self._mult_converted(x, y):
            return
        self.check_layout(x, y)
        if self._tr is not None or self._native_tr:
            if self._native_tr:
                # split and schedule inside the library (dnm_mat_set_exchange, csrc/comm.cpp)
                if self._native is None:
                    self._native = native_comm()
                _lib.check(L.dnm_mat_mult_partitioned(self.handle, self._native, x.ptr, y.ptr, _stream()))
            else:
                self._mult_transposed(x, y)
            if self._check_pending:
                # The transposed exchange is the scheme with the most asynchronous traffic (buffers shared between
                # the RCCL stream and the compute stream, batched returns): its first result on a transport is
                # checked against rows recomputed from the MSC definition before anything is built on it.
                self._check_pending = False
                err, scale = self.selfcheck(x, y)
                if not err <= 1e-9 * max(scale, 1e-300):
                    raise ExchangeCheckError('transposed exchange: sampled rows of the first multiply are off by %.3e '
                                             '(scale %.3e); build the operator with exchange="partner"' % (err, scale))
            return
        if self._native is None and self.nranks > 1 and self._native_applies(x):
            self._native = native_comm()
        if self._native is not None and self.nranks > 1:
            _lib.check(L.dnm_mat_mult_partitioned(self.handle, self._native, x.ptr, y.ptr, _stream()))
            return
        if self.nranks > 1 and not self.partners and self._is_windowed():
            return self._mult_window(x, y)
        if not self.recvs and not self.sends:
            _lib.check(L.dnm_mat_mult(self.handle, x.ptr, y.ptr, _stream()))
            return
        import torch
        bufs = []
        for i, (p, off, cnt) in enumerate(self.recvs):
            if i not in self._recv:
                self._recv[i] = torch.empty(cnt, dtype=x.array.dtype, device=x.array.device)
            bufs.append(self._recv[i])
        reqs = post_exchange(x.array, self.sends, self.recvs, bufs)     # runs on RCCL's stream
        _lib.check(L.dnm_mat_mult_local(self.handle, x.ptr, y.ptr, _stream()))   # overlaps
        # one request per posted operation (sends first, then the receives in order): wait per received block and
        # apply it at once, smallest first, so that a large block still in flight does not hold back the others;
        # a transport that hands back one request for the whole batch is waited for as a whole
        nr = len(self.recvs)
        if len(reqs) == len(self.sends) + nr:
            rreq = reqs[len(self.sends):]
            for i in sorted(range(nr), key=lambda j: self.recvs[j][2]):
                rreq[i].wait()
                _lib.check(L.dnm_mat_mult_remote(self.handle, i, C.c_void_p(bufs[i].data_ptr()), y.ptr, _stream()))
            for r in reqs[:len(self.sends)]:
                r.wait()
        else:
            for r in reqs:
                r.wait()
            for i in range(nr):
                _lib.check(L.dnm_mat_mult_remote(self.handle, i, C.c_void_p(bufs[i].data_ptr()), y.ptr, _stream()))

    def vec_in(self, v):
        """``v`` as an input vector in this matrix's layout: itself, or a copy made through the reference order (a
        state of the subspace's own layout handed to an operator that works in a relabelled one)."""
        if (v.swz, v.perm) == (self.swz_right, self.perm_right):
            return v
        if not (v.internal and self.swz_right >= 256 and self.nranks == 1):
            return v                      # check_layout names the mismatch
        out = Vec(v.size, swz=self.swz_right, sub_c=self._keep[1])
        out.set_local_natural(v.local_natural())
        return out

    def vec_out(self, v):
        """A result vector in this matrix's layout standing in for ``v`` (itself when the layouts agree); the caller
        copies it back with ``Vec.copy``, which converts."""
        if (v.swz, v.perm) == (self.swz_left, self.perm_left):
            return v
        if not (v.internal and self.swz_left >= 256 and self.nranks == 1):
            return v
        return Vec(v.size, swz=self.swz_left, sub_c=self._keep[0])

    def _native_applies(self, x):
        """The native schedule moves device memory over RCCL: every partition takes it (round 6: also window partitions whose
        right vectors are stored swizzled -- the library straightens the block before it travels); the host schedules
        remain for the gloo-staged transport of the CPU / one-GPU tests."""
        return native_transport() and x.array.is_cuda

    def check_layout(self, x, y):
        """Raise unless the vectors are laid out as this matrix expects them (x: right subspace, y: left).  Every
        caller that hands raw vector pointers to the native library -- ``mult`` and the Krylov solvers -- goes
        through here: the layout of a vector is fixed when it is created, the matrix's when it is built, and
        process state in between (``config.vec_swizzle``, the number of ranks) may have changed."""
        for v, want, wperm, n, name in ((x, self.swz_right, self.perm_right, self.n_local, 'input'),
                                        (y, self.swz_left, self.perm_left, self.m_local, 'result')):
            if v.swz != want:
                raise ValueError('%s vector layout (swizzle %d) does not match the matrix (%d): the state was '
                                 'created under a different vector layout or rank count than the operator'
                                 % (name, v.swz, want))
            if getattr(v, 'perm', None) != wperm:
                raise ValueError('%s vector and matrix differ in the site relabelling of their SpinConserve layout'
                                 % name)
            if v.local_size != n or v.array.numel() != n:
                raise ValueError('%s vector holds %d local elements (array of %d), the matrix expects %d'
                                 % (name, v.local_size, v.array.numel(), n))

    def selfcheck(self, x, y, nsample=24):
        """Collective: recompute ``nsample`` rows per rank of y = A x on the host from the operator's definition
        (A.3 of SURVEY.md: H[r, col] = sum_t (1 - 2 parity(bra & sign_t)) c_t over the terms of every mask, bra the
        column state) with the x values fetched from the ranks that own them, and return (largest absolute
        deviation over all ranks, largest |y| sampled).  Independent of every kernel and of the exchange scheme."""
        import torch
        from . import _comm
        if self._msc is None:
            raise RuntimeError('selfcheck needs the operator arrays (matrices built by build_mat have them)')
        if knob('DNM_TEST_FAIL_SELFCHECK') == '1':       # tests: what a wrong first multiply looks like to the caller
            return 1.0, 1.0
        masks, offs, signs, coeffs, lsub, rsub = self._msc
        lib = _lib.lib()
        # A real-packed handle (Full / Parity; eigsolve's real arithmetic) holds two real amplitudes per complex128
        # element: the rows sampled are REAL indices -- element index >> 1 through the vector's own swizzle, lane index
        # & 1 -- and the operator's (real) matrix elements are applied to real amplitudes (the packed
        # transposed exchange ran unguarded).
        packed = bool(self.real_packed)
        if packed and (x.swz >= 256 or y.swz >= 256):
            raise RuntimeError('selfcheck: no packed mode for the SpinConserve internal layout')

        def read(v, local_idx):
            """amplitudes of the local (real, if packed) indices of a vector"""
            idx = torch.from_numpy(np.ascontiguousarray(local_idx, dtype=np.int64)).to(v.array.device)
            if not packed:
                return v.array[v.positions(idx)].cpu().numpy()
            flat = torch.view_as_real(v.array).reshape(-1)
            return flat[2 * v.positions(idx >> 1) + (idx & 1)].cpu().numpy()
        mult = 2 if packed else 1

        def maps(sub, fn, vals):
            vals = np.ascontiguousarray(vals, dtype=np.int64)
            out = np.empty_like(vals)
            _lib.check(fn(C.byref(sub['data']), vals.size, _lib.p64(vals), _lib.p64(out)))
            return out
        nloc = mult * y.rows
        # rows: both ends, the middle and the neighbourhood of every power of two of the local index
        cand = {0, nloc - 1, nloc // 2}
        b = 1
        while b < nloc:
            cand.update((b - 1, b))
            b <<= 1
        rs = np.random.RandomState(1234 + self.rank)
        rows = sorted(cand)[:max(0, nsample // 2)]
        rows = np.unique(np.concatenate([np.asarray(rows, dtype=np.int64), rs.randint(0, nloc, size=nsample - len(rows))]))
        ystart = mult * (y.start if hasattr(y, 'start') else self.rank * y.rows)
        grow = rows + ystart
        kets = maps(lsub, lib.dnm_idx_to_state, grow)
        bras = kets[:, None] ^ np.asarray(masks, dtype=np.int64)[None, :]
        cols = maps(rsub, lib.dnm_state_to_idx, bras.reshape(-1)).reshape(bras.shape)
        # fetch x[col] from the owners
        d = _dist()
        ws = self.nranks
        owned = [tuple(mult * v for v in split_ownership(self.N, ws, q)) for q in range(ws)]
        need = np.unique(cols[cols >= 0])
        allneed = [None] * ws
        if d is not None:
            d.all_gather_object(allneed, need)
        else:
            allneed = [need]
        s0, sn = owned[self.rank]
        answers = []
        for q in range(ws):
            mine = allneed[q][(allneed[q] >= s0) & (allneed[q] < s0 + sn)]
            answers.append((mine, read(x, mine - s0)))
        allans = [None] * ws
        if d is not None:
            d.all_gather_object(allans, answers)
        else:
            allans = [answers]
        xval = {}
        for q in range(ws):
            idx, val = allans[q][self.rank]
            xval.update(zip(idx.tolist(), val.tolist()))
        tm = np.repeat(np.arange(len(masks)), np.diff(offs))
        want = np.zeros(rows.size, dtype=np.complex128)
        for i in range(rows.size):
            for m in range(len(masks)):
                c = cols[i, m]
                if c < 0:
                    continue
                t = np.flatnonzero(tm == m)
                par = np.array([bin(int(bras[i, m]) & int(signs[j])).count('1') & 1 for j in t])
                want[i] += np.sum(np.where(par == 1, -1.0, 1.0) * coeffs[t]) * xval[int(c)]
        got = read(y, rows)
        out = torch.tensor([float(np.abs(got - want).max()), float(np.abs(want).max())], dtype=torch.float64,
                           device=x.array.device)
        if d is not None:
            d.all_reduce(out, op=d.ReduceOp.MAX)
        return float(out[0]), float(out[1])

    def set_transposed(self, split, left_c, right_c, flags=0):
        """Switch the partitioned multiply to the transposed exchange (``transpose_split``): two operators with
        rank-local passes only -- the masks that flip no top spin in the state's own layout, the others in the
        redistributed layout -- and the piece list of the all-to-all between the layouts."""
        lo, hi, f = split
        if self.real_packed:
            f -= 1              # positions of the packed operator: index bit 0 is the lane, the fields sit one bit lower
        p = self.nranks.bit_length() - 1
        n = (self.n_local - 1).bit_length()
        hs = []
        try:
            for which, arrs in enumerate((lo, hi)):
                fl = flags
                if which == 1 and n - f <= 8:
                    # layout B's masks live on the p exchanged bits and the top bit: a tile [0, a) + [f, n) holds them
                    # all and leaves the bits right below f -- the top bits INSIDE a piece -- to the workgroup index,
                    # so that ranges of workgroups are contiguous sub-pieces (dnm_mat_mult_local_part)
                    tile_bits = int(knob('DNM_TILE_BITS', '12'))
                    a = tile_bits - (n - f)
                    if 2 <= a <= 9:
                        fl |= a << _lib.MAT_AMIN_SHIFT
                h = create_mat(*arrs, left_c, right_c, False, fl, self.rank, self.nranks)
                hs.append(h)
                snd, rcv = exchange_plan(h)
                if snd or rcv:
                    raise RuntimeError('transposed exchange: a pass is not rank-local')
        except Exception:
            for h in hs:
                _lib.lib().dnm_mat_destroy(h)
            raise
        pieces, own, cnt = transpose_pieces(n, p, f, self.rank)
        self._tr = (hs[0], hs[1], pieces, own, cnt)
        # sub-piece pipelining of the layout-B pass: its ranges of workgroups must be the top bits inside a piece and
        # must not read outside themselves
        top, gathers = C.c_int(), C.c_int()
        _lib.check(_lib.lib().dnm_mat_local_part_bits(hs[1], C.byref(top), C.byref(gathers)))
        sub = self.TR_SUB
        logsub = sub.bit_length() - 1
        swz = int(left_c.vec_swizzle)
        self._tr_pipe = (knob('DNM_TRANSPOSE_PIPE', '1') != '0' and top.value == f - 1 and gathers.value == 0
                         and cnt % sub == 0 and n - int(knob('DNM_TILE_BITS', '12')) >= logsub     # whole tiles per range
                         and (swz == 0 or 2 * swz - 4 <= f - logsub))      # parts keep their order in the swizzled layout

    def set_native_transposed(self):
        """The transposed exchange with the split and the schedule inside the library (dnm_mat_set_exchange +
        dnm_mat_mult_partitioned): the default on RCCL transports (config.native_comm).  Returns whether the operator splits."""
        chosen = C.c_int()
        _lib.check(_lib.lib().dnm_mat_set_exchange(self.handle, _lib.EXCHANGE_TRANSPOSE, C.byref(chosen)))
        self._native_tr = chosen.value == _lib.EXCHANGE_TRANSPOSE
        return self._native_tr

    def _transposed_parts(self):
        """(lo handle, hi handle, pieces, own, cnt) of whichever side holds the split"""
        if self._tr is not None:
            return self._tr
        lo, hi, f = C.c_void_p(), C.c_void_p(), C.c_int()
        _lib.check(_lib.lib().dnm_mat_exchange_parts(self.handle, C.byref(lo), C.byref(hi), C.byref(f)))
        p = self.nranks.bit_length() - 1
        n = (self.n_local - 1).bit_length()
        return (lo, hi) + transpose_pieces(n, p, f.value, self.rank)

    def launches_per_mult(self):
        """Kernel launches of one multiply on this rank (rank-local passes, partner passes / the pass in the
        transposed layout and the sum of its result)."""
        def count(h):
            nl = C.c_int()
            _lib.check(_lib.lib().dnm_mat_plan_launches(h, C.byref(nl)))
            return nl.value
        if self._tr is not None or self._native_tr:
            tr = self._transposed_parts()
            return count(tr[0]) + count(tr[1]) * (self.TR_SUB if (self._tr_pipe or self._native_tr) else 1) + 1
        return count(self.handle) + len(self.recvs)

    def exchange_summary(self):
        """What one multiply moves between ranks: bytes received, sent, peers, and the bytes on the busiest
        link (peer) -- for the link-bound estimate of bench.py.  On a window partition the first call sets the
        windows up, which is collective: call it on every rank."""
        if self._tr is not None or self._native_tr:
            tr = self._transposed_parts()
            pieces, cnt = tr[2], tr[4]
            per_peer = {}
            for q, _, c in pieces:
                per_peer[q] = per_peer.get(q, 0) + 2 * 16 * c        # state out and result back
            tot = sum(per_peer.values())
            return {'scheme': 'transpose', 'bytes_in': tot, 'bytes_out': tot, 'peers': len(per_peer),
                    'busiest_link_bytes': max(per_peer.values()) if per_peer else 0}
        per_peer = {}
        for q, _, c in self.recvs:
            per_peer[q] = per_peer.get(q, 0) + 16 * c
        if self.nranks > 1 and not self.partners and self._is_windowed():
            self._setup_windows()
            rcv, snd = window_exchange_ops(self._owned, self._windows, self.rank, self._needs)
            for q, lo, hi in rcv:
                per_peer[q] = per_peer.get(q, 0) + 16 * (hi - lo)
            return {'scheme': 'window', 'bytes_in': sum(per_peer.values()),
                    'bytes_out': sum(16 * (hi - lo) for _, lo, hi in snd), 'peers': len(per_peer),
                    'busiest_link_bytes': max(per_peer.values()) if per_peer else 0,
                    'window_bytes': 16 * (self._windows[self.rank][1] - self._windows[self.rank][0] + 1)}
        return {'scheme': 'window' if (self.nranks > 1 and not self.partners and self._is_windowed()) else 'partner',
                'bytes_in': sum(per_peer.values()), 'bytes_out': sum(16 * c for _, _, c in self.sends),
                'peers': len(per_peer), 'busiest_link_bytes': max(per_peer.values()) if per_peer else 0}

    def _transpose_buffers(self, like):
        import torch
        if self._tr_bufs is None:
            self._tr_bufs = (torch.empty(self.n_local, dtype=like.dtype, device=like.device),
                             torch.empty(self.n_local, dtype=like.dtype, device=like.device))
        return self._tr_bufs

    def _mult_transposed(self, x, y):
        """y = A x with the transposed exchange: the state goes to layout B (all-to-all, every link at once) while
        the masks that flip no top spin run here; the others are one rank-local pass in layout B, and its result
        comes back through the same all-to-all and is added.  ``self.trace`` (a list, tools/transpose_timeline.py):
        every phase appends (name, seconds since the call began) after the device has finished it."""
        if self._tr_pipe:
            return self._mult_transposed_pipelined(x, y)
        L = _lib.lib()
        lo, hi, pieces, own, cnt = self._tr
        xb, wb = self._transpose_buffers(x.array)
        vp = lambda t: C.c_void_p(t.data_ptr())
        mark = self._trace_marker()
        reqs = post_transpose(x.array, xb, pieces)                              # runs on RCCL's stream
        mark('forward all-to-all posted')
        for off in own:
            _lib.check(L.dnm_vec_copy(vp(x.array[off:off + cnt]), vp(xb[off:off + cnt]), cnt, _stream()))
        _lib.check(L.dnm_mat_mult_local(lo, x.ptr, y.ptr, _stream()))           # overlaps the all-to-all
        mark('layout-A passes (masks that flip no rank bit)')
        for r in reqs:
            r.wait()
        mark('forward all-to-all complete')
        _lib.check(L.dnm_mat_mult_local(hi, vp(xb), vp(wb), _stream()))
        mark('layout-B pass (masks that flip rank bits)')
        # the way back (xb is free again) in TR_SUB * len(own) batches -- every piece travels as TR_SUB contiguous parts,
        # part by part over all peers: what a batch brought is added to y while the next ones are on the links, so the
        # only addition that is not hidden under a transfer is the last batch's (1 / (TR_SUB * len(own)) of the sweep)
        span = self.n_local // len(own)               # P pieces: one per peer around this rank's own
        sub = self.TR_SUB if cnt % self.TR_SUB == 0 and cnt // self.TR_SUB >= 1024 else 1
        part = cnt // sub
        batches = []
        for b in range(len(own)):
            mine = [pc for pc in pieces if pc[1] // span == b]
            for k in range(sub):
                batches.append((b, k, post_transpose(wb, xb, [(q, off + k * part, part) for q, off, _ in mine])))
        mark('return all-to-all posted in %d batches' % len(batches))
        for off in own:
            _lib.check(L.dnm_vec_axpby(vp(y.array[off:off + cnt]), vp(wb[off:off + cnt]), cnt, 1.0, 0.0, 1.0, 0.0,
                                       _stream()))
        mark('own pieces added')
        for i, (b, k, reqs) in enumerate(batches):
            for r in reqs:
                r.wait()
            # the parts k of the pieces on either side of this rank's own piece
            for q, poff, _ in [pc for pc in pieces if pc[1] // span == b]:
                lo_ = poff + k * part
                _lib.check(L.dnm_vec_axpby(vp(y.array[lo_:lo_ + part]), vp(xb[lo_:lo_ + part]), part, 1.0, 0.0, 1.0, 0.0,
                                           _stream()))
            mark('return batch %d of %d received and added' % (i + 1, len(batches)))

    def _mult_transposed_pipelined(self, x, y):
        """The transposed exchange sub-piece by sub-piece (the reference overlaps assembly of one block with compute of
        the next, bpetsc_template_2.c:866-873).  Every piece of 2^f amplitudes is cut into TR_SUB contiguous parts; part
        s of ALL pieces is what range s of the layout-B pass's workgroups reads and writes (its tile holds the
        exchanged bits and the top bit, its workgroup index the bits below).  So: the forward all-to-all is posted part
        by part; the layout-A passes run under it; as soon as part s has landed, range s of the layout-B pass runs and
        its result goes straight back -- while part s + 1 is still arriving -- and is added to y on arrival.  Exposed
        at the end: one part's return and addition."""
        L = _lib.lib()
        lo, hi, pieces, own, cnt = self._tr
        xb, wb = self._transpose_buffers(x.array)
        vp = lambda t: C.c_void_p(t.data_ptr())
        mark = self._trace_marker()
        sub = self.TR_SUB
        part = cnt // sub
        fwd = [post_transpose(x.array, xb, [(q, off + s * part, part) for q, off, _ in pieces]) for s in range(sub)]
        mark('forward all-to-all posted in %d parts' % sub)
        for off in own:
            _lib.check(L.dnm_vec_copy(vp(x.array[off:off + cnt]), vp(xb[off:off + cnt]), cnt, _stream()))
        _lib.check(L.dnm_mat_mult_local(lo, x.ptr, y.ptr, _stream()))           # overlaps the all-to-all
        mark('layout-A passes (masks that flip no rank bit)')
        back = []
        for s in range(sub):
            for r in fwd[s]:
                r.wait()
            _lib.check(L.dnm_mat_mult_local_part(hi, vp(xb), vp(wb), s, sub, _stream()))
            # part s of xb has been consumed: it takes the returning part s
            back.append(post_transpose(wb, xb, [(q, off + s * part, part) for q, off, _ in pieces]))
            for off in own:
                o = off + s * part
                _lib.check(L.dnm_vec_axpby(vp(y.array[o:o + part]), vp(wb[o:o + part]), part, 1.0, 0.0, 1.0, 0.0, _stream()))
            mark('part %d: landed, layout-B range run, return posted, own part added' % (s + 1))
        for s in range(sub):
            for r in back[s]:
                r.wait()
            for q, off, _ in pieces:
                o = off + s * part
                _lib.check(L.dnm_vec_axpby(vp(y.array[o:o + part]), vp(xb[o:o + part]), part, 1.0, 0.0, 1.0, 0.0, _stream()))
            mark('returned part %d added' % (s + 1))

    trace = None   # set to a list to record the phases of _mult_transposed

    def _trace_marker(self):
        if self.trace is None:
            return lambda name: None
        import time
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()

        def mark(name):
            torch.cuda.synchronize()
            self.trace.append((name, time.perf_counter() - t0))
        return mark

    TR_SUB = 4     # parts a piece of the returning all-to-all travels in (_mult_transposed)

    def _native_in_use(self, x):
        """Whether ``mult`` runs this operator through dnm_mat_mult_partitioned (binds the communicator on first use)."""
        if self.nranks == 1:
            return False
        if self._native is None and (self._native_tr or (self._tr is None and self._native_applies(x))):
            self._native = native_comm()
        return self._native is not None and (self._native_tr or self._tr is None)

    def _native_phase(self, x, y, phase):
        L = _lib.lib()
        _lib.check(L.dnm_comm_set_phase(self._native, phase))
        try:
            _lib.check(L.dnm_mat_mult_partitioned(self.handle, self._native, x.ptr, y.ptr, _stream()))
        finally:
            _lib.check(L.dnm_comm_set_phase(self._native, _lib.PHASE_ALL))

    def compute_only(self, x, y):
        """The kernels of ONE multiply with nothing on the links (receive buffers / the window / the redistributed state
        hold whatever the last multiply left there): the rank's compute time, for bench.py's split of a partitioned
        multiply into exchange, compute and what the schedule hides.  ``y`` does not hold A x afterwards."""
        L = _lib.lib()
        if self.nranks == 1:
            _lib.check(L.dnm_mat_mult(self.handle, x.ptr, y.ptr, _stream()))
            return
        if self._native_in_use(x):
            return self._native_phase(x, y, _lib.PHASE_COMPUTE)
        vp = lambda t: C.c_void_p(t.data_ptr())
        if self._tr is not None:
            lo, hi, pieces, own, cnt = self._tr
            xb, wb = self._transpose_buffers(x.array)
            _lib.check(L.dnm_mat_mult_local(lo, x.ptr, y.ptr, _stream()))
            _lib.check(L.dnm_mat_mult_local(hi, vp(xb), vp(wb), _stream()))
            _lib.check(L.dnm_vec_axpby(y.ptr, vp(wb), self.n_local, 1.0, 0.0, 1.0, 0.0, _stream()))
        elif not self.partners and self._is_windowed():
            self._setup_windows()
            self.prepare_exchange(x.array)
            w0, wb = self._windows[self.rank][0], self._window_buf
            if self._window_splits():
                _lib.check(L.dnm_mat_mult_window_local(self.handle, x.ptr, y.ptr, _stream()))
                _lib.check(L.dnm_mat_mult_window_remote(self.handle, vp(wb), w0, wb.numel(), y.ptr, _stream()))
            else:
                _lib.check(L.dnm_mat_mult_window(self.handle, vp(wb), w0, wb.numel(), y.ptr, _stream()))
        else:
            self.prepare_exchange(x.array)
            _lib.check(L.dnm_mat_mult_local(self.handle, x.ptr, y.ptr, _stream()))
            for i in range(len(self.recvs)):
                _lib.check(L.dnm_mat_mult_remote(self.handle, i, vp(self._recv[i]), y.ptr, _stream()))

    def exchange_only(self, x, y=None):
        """Post and complete the rank exchange of ONE multiply without running any kernel: the same messages over
        the same transport, for measuring what the links sustain (bench.py's ``xgmi_link_GBs_measured``).
        Collective: every rank calls it.  Under the native schedule (``y``: any result vector; it is left alone) the
        library posts the very groups of its multiply (dnm_comm_set_phase)."""
        if self.nranks == 1:
            return
        import torch
        if y is not None and self._native_in_use(x):
            self._native_phase(x, y, _lib.PHASE_EXCHANGE)
            torch.cuda.synchronize()
            return
        if self._tr is not None or self._native_tr:
            _, _, pieces, own, cnt = self._transposed_parts()
            xb, wb = self._transpose_buffers(x.array)
            for r in post_transpose(x.array, xb, pieces):        # the state goes out ...
                r.wait()
            for r in post_transpose(wb, xb, pieces):             # ... and a result of the same size comes back
                r.wait()
            if self._native_tr:
                self._tr_bufs = None                             # (the library has its own pair)
        elif not self.partners and self._is_windowed():
            self._setup_windows()
            self._window_buf = exchange_window(x.array if x.internal else x.local_natural(), self._owned, self._windows,
                                               self.rank, self._window_buf, self._needs)
        else:
            self.prepare_exchange(x.array)
            bufs = [self._recv[i] for i in range(len(self.recvs))]
            for r in post_exchange(x.array, self.sends, self.recvs, bufs):
                r.wait()
        if x.array.is_cuda:
            torch.cuda.synchronize()

    def prepare_exchange(self, like):
        """Allocate the receive buffers / column window of the partitioned multiply now (they are
        otherwise created by the first ``mult``), so that a solver sizing its Krylov basis to the
        free device memory sees what is really left.  ``like``: a local vector (dtype / device)."""
        if self.nranks == 1:
            return
        import torch
        if self._native_tr or self._native is not None:
            if self._native is None:
                self._native = native_comm()
            _lib.check(_lib.lib().dnm_comm_prepare(self._native, self.handle, _stream()))
            return
        if self._tr is not None:
            self._transpose_buffers(like)
            return
        if not self.partners and self._is_windowed():
            self._setup_windows()
            lo, hi = self._windows[self.rank]
            if self._window_buf is None or self._window_buf.numel() != hi - lo + 1:
                self._window_buf = torch.zeros(hi - lo + 1, dtype=like.dtype, device=like.device)
            return
        for i, (p, off, cnt) in enumerate(self.recvs):
            if i not in self._recv:
                self._recv[i] = torch.empty(cnt, dtype=like.dtype, device=like.device)

    def uses_cached_diagonal(self):
        """True for the kernels that read a cached diagonal: the SpinConserve kernel (also
        partitioned) and the generic row-gather kernel on one rank."""
        d = self.describe()
        if 'SpinConserve kernel' in d or 'SpinConserve row kernel' in d or 'diagonal cached' in d:
            return True
        return 'row-gather kernel' in d

    def _window_splits(self):
        if self._splits is None:
            v = C.c_int()
            _lib.check(_lib.lib().dnm_mat_window_split(self.handle, C.byref(v)))
            self._splits = bool(v.value)
        return self._splits

    def _is_windowed(self):
        """Partitions other than Full/Parity on 2^p ranks: rows split in index order (PetscSplitOwnership), the
        columns a rank reads come through a window gathered from its neighbours."""
        return 'tiled=1' not in self.describe()

    def column_window(self):
        lo, hi = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().dnm_mat_column_window(self.handle, C.byref(lo), C.byref(hi), _stream()))
        return lo.value, hi.value

    WINDOW_CHUNKS = 1024          # resolution of the needed-columns map over a rank's window

    def column_needs(self, window):
        """The [lo, hi) ranges of the window this rank's rows read (one device sweep), at a resolution of
        ``WINDOW_CHUNKS`` chunks over the window."""
        lo, hi = window
        # exactly, where the library knows it without a sweep (SpinConserve in the internal layout: whole blocks)
        n = C.c_int64()
        _lib.check(_lib.lib().dnm_mat_column_ranges(self.handle, 0, None, C.byref(n)))
        if 0 < n.value <= self.WINDOW_CHUNKS:
            rg = (C.c_int64 * (2 * n.value))()
            _lib.check(_lib.lib().dnm_mat_column_ranges(self.handle, n.value, rg, C.byref(n)))
            return [(int(rg[2 * i]), int(rg[2 * i + 1])) for i in range(n.value)]
        shift = max(0, int(hi - lo + 1).bit_length() - self.WINDOW_CHUNKS.bit_length())
        n = (hi >> shift) - (lo >> shift) + 1
        cmap = np.zeros(n, dtype=np.uint8)
        _lib.check(_lib.lib().dnm_mat_column_chunks(self.handle, shift, cmap.ctypes.data_as(C.POINTER(C.c_uint8)), n,
                                                    _stream()))
        return needed_ranges(cmap, shift, window)

    def _setup_windows(self):
        """Every rank's column window and, inside it, the ranges it really reads (DNM_WINDOW_RANGES=0: the whole
        window travels)."""
        if self._windows is not None:
            return
        import os
        import torch.distributed as dist
        mine = self.column_window()
        needs = self.column_needs(mine) if knob('DNM_WINDOW_RANGES', '1') != '0' else None
        allw = [None] * self.nranks
        dist.all_gather_object(allw, (mine, needs))
        self._windows = [w for w, _ in allw]
        self._needs = [nd for _, nd in allw] if needs is not None else None
        if self.swz_right >= 256:      # internal SpinConserve layout: ownership and windows are positions of the layout
            self._owned = [layout_partition(self._keep[1], self.nranks, q)[:2] for q in range(self.nranks)]
            if self.real_packed:       # real vectors: the handle counts pairs of positions (complex128 elements)
                self._owned = [(a // 2, b // 2) for a, b in self._owned]
        else:
            self._owned = [split_ownership(self.N, self.nranks, q) for q in range(self.nranks)]

    def _mult_window(self, x, y):
        """Partitioned SpinConserve: gather the column window, then one kernel."""
        self._setup_windows()
        # the window is in index order: a swizzled block is straightened first (projection pairs); vectors in the
        # internal SpinConserve layout travel as they lie (the window is a range of the layout)
        xl = x.array if x.internal else x.local_natural()
        w0 = self._windows[self.rank][0]
        if self._window_splits():
            # two tiled passes in the internal SpinConserve layout: the part that reads only what the rank owns (bonds
            # inside a block of equal top bits, the diagonal) runs while the window is on the links, the rest adds to
            # it once the window is complete (the reference overlaps assembly and compute block by block,
            # bpetsc_template_2.c:866-873)
            L = _lib.lib()
            self._window_buf, reqs = post_window_exchange(xl, self._owned, self._windows, self.rank, self._window_buf,
                                                          self._needs)
            _lib.check(L.dnm_mat_mult_window_local(self.handle, x.ptr, y.ptr, _stream()))
            for r in reqs:
                r.wait()
            _lib.check(L.dnm_mat_mult_window_remote(self.handle, C.c_void_p(self._window_buf.data_ptr()), w0,
                                                    self._window_buf.numel(), y.ptr, _stream()))
            return
        local, remote = self._window_row_ranges()
        if local:
            # reference order, Explicit, projections, odd rank counts: the rows that read only the rank's own block of x
            # (whole stretches of equal top bits) are multiplied while the window is on the links, the others after
            L = _lib.lib()
            self._window_buf, reqs = post_window_exchange(xl, self._owned, self._windows, self.rank, self._window_buf,
                                                          self._needs)
            wp, wn = C.c_void_p(self._window_buf.data_ptr()), self._window_buf.numel()
            for r0, r1 in local:
                _lib.check(L.dnm_mat_mult_window_rows(self.handle, wp, w0, wn, y.ptr, r0, r1, _stream()))
            for r in reqs:
                r.wait()
            for r0, r1 in remote:
                _lib.check(L.dnm_mat_mult_window_rows(self.handle, wp, w0, wn, y.ptr, r0, r1, _stream()))
            return
        self._window_buf = exchange_window(xl, self._owned, self._windows, self.rank, self._window_buf, self._needs)
        _lib.check(_lib.lib().dnm_mat_mult_window(self.handle, C.c_void_p(self._window_buf.data_ptr()), w0,
                                                  self._window_buf.numel(), y.ptr, _stream()))

    WINDOW_ROW_RANGES = 8         # row ranges multiplied under the window exchange, at most
    WINDOW_ROWS_MIN_SHARE = 0.05  # ... if they are at least this share of the rank's rows
    WINDOW_ROWS_MIN_BLOCKS = 64   # ... and no range shorter than this many workgroups of 256 rows

    def _window_row_ranges(self):
        """(local, remote): ranges [r0, r1) of this rank's rows that read nothing but its own block of x, and the
        rest (dnm_mat_window_local_rows; one sweep, cached).  ([], []) when the multiply does not split that way."""
        if self._row_ranges is None:
            self._row_ranges = ([], [])
            my0, myn = self._owned[self.rank]
            buf = (C.c_int64 * (2 * self.WINDOW_ROW_RANGES))()
            n = C.c_int()
            _lib.check(_lib.lib().dnm_mat_window_local_rows(self.handle, my0, my0 + myn, self.WINDOW_ROW_RANGES,
                                                            self.WINDOW_ROWS_MIN_BLOCKS, buf, C.byref(n), _stream()))
            local = [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n.value)]
            if knob('DNM_WINDOW_ROWS', '1') != '0' and sum(b - a for a, b in local) >= self.WINDOW_ROWS_MIN_SHARE * self.m_local:
                self._row_ranges = (local, complement_ranges(local, self.m_local))
        return self._row_ranges

    def norm(self, norm_type='infinity'):
        if norm_type not in ('infinity', None):
            raise ValueError('Only NORM_INFINITY is implemented for shell matrices.')
        v = C.c_double()
        _lib.check(_lib.lib().dnm_mat_norm_inf(self.handle, C.byref(v), _stream()))
        d = _dist()
        if d is not None:
            import torch
            t = torch.tensor([v.value], dtype=torch.float64, device=config.device)
            d.all_reduce(t, op=d.ReduceOp.MAX)
            v.value = float(t.item())
            _lib.check(_lib.lib().dnm_mat_set_norm(self.handle, v.value))
        return v.value

    def precompute_diagonal(self):
        _lib.check(_lib.lib().dnm_mat_precompute_diagonal(self.handle, _stream()))

    def destroy(self):
        if self._h is not None:
            if self._native is not None:
                # (a communicator released before its operators -- release_native_comm -- has nothing left to forget)
                if _NATIVE_COMM is not None and self._native.value == _NATIVE_COMM.value:
                    _lib.check(_lib.lib().dnm_comm_forget(self._native, self._h))
                self._native = None
            _lib.check(_lib.lib().dnm_mat_destroy(self._h))
            self._h = None
            self._recv = {}
            if self._tr is not None:
                for h in self._tr[:2]:
                    _lib.check(_lib.lib().dnm_mat_destroy(h))
                self._tr, self._tr_bufs = None, None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def create_mat(masks, mask_offsets, signs, coeffs, left_c, right_c, xparity=False, flags=0,
               rank=0, nranks=1):
    """Raw dnm_mat_create; left_c / right_c are ctypes dnm_subspace structs."""
    masks = np.ascontiguousarray(masks, dtype=np.int64)
    mask_offsets = np.ascontiguousarray(mask_offsets, dtype=np.int64)
    signs = np.ascontiguousarray(signs, dtype=np.int64)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.complex128)
    part = _lib.Partition(rank, nranks)
    h = C.c_void_p()
    _lib.check(_lib.lib().dnm_mat_create(
        masks.size, _lib.p64(masks), _lib.p64(mask_offsets), _lib.p64(signs),
        coeffs.view(np.float64).ctypes.data_as(_lib.f64p), C.byref(left_c), C.byref(right_c),
        int(bool(xparity)), int(flags), C.byref(part), C.byref(h)))
    return h


def _relabelled(masks, lc, rc, xparity, site_perm):
    """The descriptors a SpinConserve pair in the internal layout is built on: with a site relabelling
    (dnm_subspace.site_perm) when the operator's bond graph gains from one -- ``site_perm``: None = choose
    (dnm_sc_choose_site_perm; the identity for chains), False = never, an array = that one.  One rank, same subspace on
    both sides (XParity on top of it: spin L-1 stays); everything else keeps the descriptors as they are."""
    if site_perm is False or config.world_size != 1 or not config.sc_site_perm:
        return lc, rc
    if xparity and 2 * int(lc.k) != int(lc.L):
        return lc, rc
    if not (lc.type == 3 and rc.type == 3 and lc.L == rc.L and lc.k == rc.k and lc.vec_swizzle >= 256
            and lc.vec_swizzle == rc.vec_swizzle and not lc.site_perm and not rc.site_perm):
        return lc, rc
    L = int(lc.L)
    if site_perm is None:
        a, w = lc.vec_swizzle & 0xff, (lc.vec_swizzle >> 8) & 0xff
        # (XParity on top: spin L-1 -- the one that tells a representative from its mirror image -- keeps its place)
        site_perm, _ = choose_site_perm(np.unique(np.asarray(masks, dtype=np.int64)), L, a, w, fix_top=bool(xparity))
    site_perm = np.ascontiguousarray(site_perm, dtype=np.int8)
    if np.array_equal(site_perm, np.arange(L)):
        return lc, rc
    d = with_site_perm(lc, site_perm)
    return d, d


def build_mat(masks, mask_offsets, signs, coeffs, left_subspace, right_subspace, xparity=False,
              shell=True, gpu=True, flags=0, exchange=None, site_perm=None):
    """Mirror of ``bpetsc.build_mat`` (bpetsc.pyx:78-138).  ``left_subspace`` /
    ``right_subspace`` are the dicts ``Subspace._to_c()`` returns.  ``exchange`` (not in the reference):
    'partner' / 'transpose' picks the scheme of a partitioned Full / Parity multiply, None decides by rank count.
    ``site_perm`` (not in the reference): see ``_relabelled``."""
    if not shell:
        raise ValueError('this engine builds matrix-free (shell) operators only')
    if not gpu:
        raise RuntimeError('dynamite_amd has no CPU path')
    config._initialize()
    lc, rc = left_subspace['data'], right_subspace['data']
    lc, rc = _relabelled(masks, lc, rc, xparity, site_perm)
    h = create_mat(masks, mask_offsets, signs, coeffs, lc, rc, xparity, flags,
                   rank=config.rank, nranks=config.world_size)
    mat = ShellMat(h, lc, rc, config.world_size, config.rank)
    if not xparity:
        mat._msc = (np.array(masks, dtype=np.int64), np.array(mask_offsets, dtype=np.int64),
                    np.array(signs, dtype=np.int64), np.array(coeffs, dtype=np.complex128), left_subspace, right_subspace)
    same = lc.type == rc.type and lc.L == rc.L and (lc.type == 0 or (lc.type == 1 and lc.space == rc.space))
    if use_transposed_exchange(config.world_size, exchange) and same and not xparity and 'tiled=1' in mat.describe():
        shift = int(lc.type)                   # Full: index = configuration; Parity: index = configuration >> 1
        split = transpose_split(masks, mask_offsets, signs, coeffs, int(lc.L) - shift, config.world_size,
                                int(lc.vec_swizzle), shift, packed=mat.real_packed)
        if split is not None and native_transport() and mat.set_native_transposed():
            mat._check_pending = knob('DNM_EXCHANGE_SELFCHECK', '1') != '0'
        elif split is not None:
            mat.set_transposed(split, lc, rc, flags)
            # (a packed operator is checked on real amplitudes: ShellMat.selfcheck's packed mode)
            mat._check_pending = knob('DNM_EXCHANGE_SELFCHECK', '1') != '0'
    return mat


def use_transposed_exchange(nranks, mode=None):
    """Exchange scheme of a partitioned Full-space or Parity multiply: with four or more ranks the all-to-all of the
    transposed scheme puts less on the busiest link than the partner blocks (two ranks: the partner block is
    half of what two transposes move).  DNM_EXCHANGE=partner / transpose overrides (an experiment knob: counted only
    under DNM_EXPERIMENTAL=1, like every DNM_* variable; `build_mat(..., exchange=...)` is the production way)."""
    mode = mode or knob('DNM_EXCHANGE', 'auto')
    if mode == 'partner' or nranks < 2:
        return False
    return mode == 'transpose' or nranks >= 4


def check_conserves(masks, mask_offsets, signs, coeffs, left_subspace, right_subspace, xparity=False):
    """Mirror of ``bpetsc.check_conserves`` (bpetsc.pyx:150-193), run on the GPU."""
    config._initialize()
    masks = np.ascontiguousarray(masks, dtype=np.int64)
    mask_offsets = np.ascontiguousarray(mask_offsets, dtype=np.int64)
    signs = np.ascontiguousarray(signs, dtype=np.int64)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.complex128)
    res = C.c_int()
    _lib.check(_lib.lib().dnm_check_conserves(
        masks.size, _lib.p64(masks), _lib.p64(mask_offsets), _lib.p64(signs),
        coeffs.view(np.float64).ctypes.data_as(_lib.f64p), C.byref(left_subspace['data']),
        C.byref(right_subspace['data']), int(bool(xparity)), C.byref(res), _stream()))
    return bool(res.value)


def rdm_block_subspace(subspace, rank, nranks, keep):
    """The subspace a rank's block of a partitioned state lives on, as far as the reduced density
    matrix of ``keep`` is concerned -- or None if the kept spins are not all inside one block.
    Full(L) on P = 2^p ranks: the block is a Full(L - p) vector (the top p spins are the rank).
    Parity(L, s): basis index = configuration >> 1, so the block fixes the top p spins and is a
    Parity(L - p, s ^ parity(rank)) vector.  Other subspaces are not cut along spins."""
    d = subspace['data']
    p = nranks.bit_length() - 1
    if nranks != 1 << p or subspace['type'] not in (0, 1):      # FULL, PARITY
        return None
    Lb = int(d.L) - p
    if Lb < 1 or (len(keep) and int(max(keep)) >= Lb):
        return None
    out = _lib.Subspace()
    out.type, out.L = d.type, Lb
    out.vec_swizzle = d.vec_swizzle        # the swizzle acts on the local index
    if subspace['type'] == 1:
        out.space = int(d.space) ^ (bin(rank).count('1') & 1)
    return out


def rdm_partial(x_block, sub_c, keep):
    """Device tensor (4^len(keep) complex128): sum over the traced configurations of one block."""
    import torch
    K = 1 << keep.size
    rho = torch.empty(K * K, dtype=torch.complex128, device=x_block.device)
    _lib.check(_lib.lib().dnm_reduced_density_matrix(
        C.c_void_p(x_block.data_ptr()), C.byref(sub_c), keep.size, _lib.p64(keep),
        C.c_void_p(rho.data_ptr()), _stream()))
    return rho


def reduced_density_matrix(vec, subspace, keep, on_device=False):
    """Mirror of ``bpetsc.reduced_density_matrix`` (bpetsc.pyx:245-276): host array of
    shape (2^len(keep),)*2 on rank 0, ``[[-1]]`` elsewhere.  Partitioned states: when the kept
    spins lie inside every rank's block (Full / Parity), each rank sums over its own traced
    configurations and the partial matrices are added on rank 0; otherwise the state is gathered on
    rank 0 as the reference does (bpetsc_template_1.c:126-141).
    ``on_device``: the matrix stays in HBM -- a (2^k, 2^k) complex128 device tensor on rank 0, None elsewhere (the
    entropies diagonalise it there: a 8192 x 8192 matrix is 1 GiB to copy and minutes of host LAPACK)."""
    rho = _reduced_density_matrix(vec, subspace, keep)
    K = 1 << len(keep)
    if rho is None:
        return None if on_device else np.array([[-1]], dtype=np.complex128)
    return rho.reshape(K, K) if on_device else rho.cpu().numpy().reshape(K, K)


def _reduced_density_matrix(vec, subspace, keep):
    """The reduced density matrix as a flat device tensor on rank 0, None on the other ranks."""
    import torch
    config._initialize()
    keep = np.ascontiguousarray(keep, dtype=np.int64)
    d = _dist()
    x = vec.array
    K = 1 << keep.size
    if d is not None and d.get_world_size() > 1:
        blk = rdm_block_subspace(subspace, config.rank, config.world_size, keep)
        if blk is not None:
            from . import _comm
            rho = rdm_partial(x, blk, keep)
            _comm.reduce_sum(rho, dst=0)
            return rho if config.rank == 0 else None
        from . import _comm
        if vec.internal:
            sizes = [layout_partition(vec.sub_c, config.world_size, q)[3] for q in range(config.world_size)]
        else:
            sizes = [split_ownership(vec.size, config.world_size, q)[1] for q in range(config.world_size)]
        parts = _comm.gather_varied(vec.local_natural(), sizes, dst=0)
        if config.rank != 0:
            return None
        x = torch.cat(parts)                 # the whole state in index order
        sub_c = _lib.Subspace.from_buffer_copy(subspace['data'])
        sub_c.vec_swizzle = 0
        return rdm_partial(x, sub_c, keep)
    sub_c = subspace['data']
    if vec.internal:
        # the kernel gathers by reference index: hand it the state in that order
        x = vec.local_natural()
        sub_c = _lib.Subspace.from_buffer_copy(sub_c)
        sub_c.vec_swizzle = 0
    return rdm_partial(x, sub_c, keep)


def precompute_diagonal(mat):
    """Mirror of ``bpetsc.precompute_diagonal`` (bpetsc.pyx:141-147)."""
    mat.precompute_diagonal()
