"""
Test-only numpy emulation of ``tile_pass_kernel`` (dynamite_amd/csrc/
matvec_kernels.hip), driven by the REAL pass tables the product's host code
builds (exported through ``dnm_mat_export_pass`` from a DNM_MAT_HOST_ONLY
handle).  It lets the CPU suite check the planner, the tile geometry and the
term tables against the oracle without a GPU.  It is not a fallback: nothing in
dynamite_amd imports it.
"""
import ctypes as C

import numpy as np

from dynamite_amd import _lib, backend


def _popc(v):
    v = np.asarray(v, dtype=np.uint64).copy()
    c = np.zeros(v.shape, dtype=np.int64)
    while np.any(v):
        c += (v & np.uint64(1)).astype(np.int64)
        v >>= np.uint64(1)
    return c


class _Quads(list):
    """The records of a pass; ``dtile``: its tabulated in-tile diagonal (DevPass::dtile), if any; ``tabs`` / ``tabvals``:
    its table records (plan.h: DevTab) and their tables as complex numbers."""
    dtile = None
    tabs = ()
    tabvals = None


class HostMat:
    """DNM_MAT_HOST_ONLY handle + exported pass tables."""

    def __init__(self, masks, mask_offsets, signs, coeffs, left_c, right_c, rank=0, nranks=1, flags=0):
        self._keep = (left_c, right_c)
        self.h = backend.create_mat(masks, mask_offsets, signs, coeffs, left_c, right_c, False,
                                    flags | _lib.MAT_HOST_ONLY, rank, nranks)
        L = _lib.lib()
        vals = [C.c_int() for _ in range(6)]
        _lib.check(L.dnm_mat_plan_counts(self.h, *[C.byref(v) for v in vals]))
        (self.n_local_passes, self.n_remote_passes, self.tiled, self.B, self.logR, self.n_loc) = \
            [v.value for v in vals]
        self.sends, self.recvs = backend.exchange_plan(self.h)
        self.partners = sorted({r[0] for r in self.recvs})
        self.local = [self._export(0, i) for i in range(self.n_local_passes)]
        self.remote = [self._export(1, i) for i in range(self.n_remote_passes)]

    def describe(self):
        buf = C.create_string_buffer(8192)
        _lib.check(_lib.lib().dnm_mat_plan_describe(self.h, buf, len(buf)))
        return buf.value.decode()

    def _export(self, remote, idx):
        L = _lib.lib()
        nq = C.c_int()
        _lib.check(L.dnm_mat_export_pass(self.h, remote, idx, None, 0, None, 0, 0, C.byref(nq)))
        desc = _lib.DevPass()
        quads = (_lib.DevQuad * max(1, nq.value))()
        _lib.check(L.dnm_mat_export_pass(self.h, remote, idx, C.byref(desc), C.sizeof(desc), quads,
                                         C.sizeof(_lib.DevQuad), nq.value, C.byref(nq)))
        out = _Quads(quads[i] for i in range(nq.value))
        nt, nv = C.c_int(), C.c_int64()
        _lib.check(L.dnm_mat_export_tabs(self.h, remote, idx, None, 0, 0, C.byref(nt), None, 0, C.byref(nv)))
        assert nt.value == desc.tab_loop[2] and desc.tab_loop[0] == 0
        if nt.value:
            tabs = (_lib.DevTab * nt.value)()
            vals = np.empty(nv.value, dtype=np.float64)
            _lib.check(L.dnm_mat_export_tabs(self.h, remote, idx, tabs, C.sizeof(_lib.DevTab), nt.value, C.byref(nt),
                                             vals.ctypes.data_as(_lib.f64p), vals.size, C.byref(nv)))
            out.tabs = [tabs[i] for i in range(nt.value)]
            out.tabvals = vals[0::2] + 1j * vals[1::2]
        if desc.has_diag:
            tab = np.empty(1 << desc.tile_bits, dtype=np.float64)
            rc = L.dnm_mat_export_dtile(self.h, remote, idx, tab.ctypes.data_as(_lib.f64p), tab.size)
            out.dtile = tab if rc == 0 else None
        return desc, out

    def __del__(self):
        try:
            _lib.lib().dnm_mat_destroy(self.h)
        except Exception:
            pass


def vec_pos(i, S):
    """Position of element i in the XOR-swizzled vector layout (dnm_subspace.vec_swizzle = S)."""
    i = np.asarray(i, dtype=np.int64)
    return i ^ (((i >> S) & ((1 << (S - 4)) - 1)) << 4) if S else i


def run_pass(hm, p, x, y, xr=None):
    """Apply one exported pass to the local vector x (numpy), updating y.  The arrays are device images: with a
    swizzled layout (desc.swz_shift) element i of x sits at vec_pos(i), of y at vec_pos(i) ^ swz_xor_y and of the
    partner slice xr at vec_pos(i) ^ swz_xor_src -- exactly the addresses the kernels form."""
    desc, quads = p
    if desc.swz_shift:
        n = 1 << desc.n_eff
        pos = vec_pos(np.arange(n), desc.swz_shift)
        assert np.array_equal(np.sort(pos), np.arange(n)) and desc.swz_xor_y < n and desc.swz_xor_src < n
        nat = type(desc).from_buffer_copy(desc)
        nat.swz_shift = 0
        ylog = y[pos ^ desc.swz_xor_y].copy()
        run_pass(hm, (nat, quads), x[pos] if desc.need_tile or xr is None else x,
                 ylog, None if xr is None else xr[pos ^ desc.swz_xor_src])
        y[pos ^ desc.swz_xor_y] = ylog
        return
    B, logR, n_loc = desc.tile_bits, desc.log_rows, desc.n_eff
    assert x.shape[0] == 1 << n_loc and y.shape[0] == 1 << n_loc
    lognt = B - logR
    NT = 1 << lognt
    R = 1 << logR
    n = 1 << n_loc
    rows = np.arange(n, dtype=np.uint64)

    tile_bits = np.uint64(0)
    for j in range(desc.nseg):
        tile_bits |= np.uint64(((1 << desc.seg_len[j]) - 1) << desc.seg_pos[j])
    # geometry self-checks (what the kernel's deposit() relies on)
    assert sum(desc.seg_len[j] for j in range(desc.nseg)) == B
    blk_bits = 0
    for j in range(desc.nbseg):
        blk_bits |= ((1 << desc.bseg_len[j]) - 1) << desc.bseg_pos[j]
    assert blk_bits & int(tile_bits) == 0 and (blk_bits | int(tile_bits)) == n - 1
    assert sum(desc.bseg_len[j] for j in range(desc.nbseg)) == n_loc - B
    offs = sorted((desc.bseg_off[j], desc.bseg_len[j]) for j in range(desc.nbseg))
    assert all(offs[i][0] + offs[i][1] == offs[i + 1][0] for i in range(len(offs) - 1))

    def compress(v):
        out = np.zeros(v.shape, dtype=np.uint64)
        for j in range(desc.nseg):
            seg = (v >> np.uint64(desc.seg_pos[j])) & np.uint64((1 << desc.seg_len[j]) - 1)
            out |= seg << np.uint64(desc.seg_off[j])
        return out

    def deposit(t):
        out = np.zeros(t.shape, dtype=np.uint64)
        for j in range(desc.nseg):
            seg = (t >> np.uint64(desc.seg_off[j])) & np.uint64((1 << desc.seg_len[j]) - 1)
            out |= seg << np.uint64(desc.seg_pos[j])
        return out

    tt = compress(rows)
    base = rows & ~tile_bits
    assert np.array_equal(base | deposit(tt), rows)
    sbase = base | np.uint64(desc.sign_base)
    tid = tt & np.uint64(NT - 1)
    kk = tt >> np.uint64(lognt)

    def amp(Q, j, tcoord):
        """slot_amp(): coeff * (-1)^(popc(tcoord & sign_tile) + popc(sbase & sign_ext))"""
        par = (_popc(tcoord & np.uint64(Q.sign_tile[j])) + _popc(sbase & np.uint64(Q.sign_ext[j]))) & 1
        return np.where(par == 1, -Q.coeff[j], Q.coeff[j])

    def kflip(a, Q, j):
        """the per-owned-row sign of a k-variant slot"""
        par = _popc(kk & np.uint64(Q.sign_tile[j] >> lognt)) & 1
        return np.where(par == 1, -a, a)

    acc = y.copy() if desc.accumulate else np.zeros(n, dtype=np.complex128)
    zero = np.zeros(n, dtype=np.uint64)

    if desc.has_diag:
        D = [np.zeros(n) for _ in range(R)]
        for q in range(desc.dext_begin, desc.dext_end):
            assert 1 <= quads[q].nslots <= 4
            for j in range(4):
                assert quads[q].sign_tile[j] == 0
                assert j < quads[q].nslots or quads[q].coeff[j] == 0
                D[0] = D[0] + amp(quads[q], j, zero)
        for b in range(R):
            for q in range(desc.dbucket[b], desc.dbucket[b + 1]):
                assert 1 <= quads[q].nslots <= 4
                for j in range(quads[q].nslots):
                    if quads[q].coeff[j] != 0:
                        assert quads[q].sign_tile[j] != 0 and (quads[q].sign_tile[j] >> lognt) == b
                    D[b] = D[b] + amp(quads[q], j, tid)
        # grouped terms (DevPass::gbucket): the group's sum over the bits outside the tile -- the same for a whole workgroup --
        # times the sign of its in-tile mask
        g0 = desc.gbucket[0]
        assert desc.gbucket[R] - g0 <= 64 and desc.gbucket[R] == desc.gbucket[_lib.MAXR]
        for b in range(R):
            for q in range(desc.gbucket[b], desc.gbucket[b + 1]):
                G = quads[q]
                assert G.nslots == 1 and G.sign_tile[0] != 0 and (G.sign_tile[0] >> lognt) == b and G.src >= 1
                Cg = np.zeros(n)
                for tq in range(G.mask_loc, G.mask_loc + G.src):
                    assert not (desc.dext_begin <= tq < desc.dext_end) and tq < g0
                    for j in range(quads[tq].nslots):
                        assert quads[tq].sign_tile[j] == 0 and quads[tq].sign_ext[j] != 0
                        Cg = Cg + amp(quads[tq], j, zero)
                D[b] = D[b] + np.where(_popc(tid & np.uint64(G.sign_tile[0])) & 1, -Cg, Cg)
        d = np.zeros(n, dtype=np.float64)      # Walsh-Hadamard over the k bits
        if getattr(quads, "dtile", None) is not None:     # tile-only terms, tabulated per tile coordinate
            assert quads.dtile.shape == (1 << B,)
            d += quads.dtile[tt.astype(np.int64)]
        for b in range(R):
            d += np.where(_popc(kk & np.uint64(b)) & 1, -D[b], D[b])
        acc += d * x

    for lp in range(_lib.LP_COUNT):
        kvar = lp in _lib.LP_KVAR
        cplx = lp in _lib.LP_CPLX
        gather = lp in _lib.LP_GATHER
        for q in range(desc.loop[lp], desc.loop[lp + 1]):
            Q = quads[q]
            if not cplx:
                assert Q.coeff[2] == 0 and Q.coeff[3] == 0
            if lp == 0:
                assert (Q.mask_tile >> lognt) == 0
            a = [amp(Q, j, tid) for j in range(4)]
            if kvar:
                a = [kflip(a[j], Q, j) for j in range(4)]
            else:
                for j in range(4):
                    assert Q.coeff[j] == 0 or (Q.sign_tile[j] >> lognt) == 0
            cre, cim = a[0] + a[1], a[2] + a[3]
            if gather:
                src = xr if Q.src else x
                xv = src[(rows ^ np.uint64(Q.mask_loc)).astype(np.int64)]
            else:
                assert desc.need_tile
                partner = base | deposit(tt ^ np.uint64(Q.mask_tile))
                xv = x[partner.astype(np.int64)]
            if desc.cache_policy & 256:
                # real-packed records (accum_record<PACK>): slots 0,1 = coefficient of the element's first lane,
                # slots 2,3 of its second; nslots != 0: a lane reads the partner element's other lane
                # (a record without lane-1 slots -- bit 0 neither flipped nor seen -- is an ordinary real record)
                assert Q.nslots in (0, 1) and (cplx or Q.nslots == 0)
                if cplx:
                    acc += cre * (xv.imag if Q.nslots else xv.real) + 1j * (cim * (xv.real if Q.nslots else xv.imag))
                else:
                    acc += cre * xv
            else:
                acc += (cre + 1j * cim) * xv

    # table records (apply_tabs): coefficient = (-1)^popc(row & z) * table[the row's bits at the flipped positions]
    chain_key = None                      # the groups of one mask: consecutive records that share the partner fetch
    for q, T in enumerate(quads.tabs):
        gather = q >= desc.tab_loop[1]
        assert 1 <= T.nbits <= 4 and not (desc.cache_policy & 256)
        # the table index as apply_tabs forms it: thread part (bit-field extracts of tid), block part (bits of the global
        # row index), k part (DevTab::ik, a nibble per row of a thread)
        idx = np.zeros(n, dtype=np.int64)
        seen = 0
        for b in range(4):
            tpos, twid = (T.tpos >> (8 * b)) & 0xff, (T.twid >> (8 * b)) & 0xff
            epos, ewid = (T.epos >> (8 * b)) & 0xff, (T.ewid >> (8 * b)) & 0xff
            assert twid in (0, 1) and ewid in (0, 1) and twid + ewid <= 1
            assert b < T.nbits or (twid, ewid) == (0, 0)
            if twid:
                assert tpos < lognt
                idx |= ((tid >> np.uint64(tpos)) & np.uint64(1)).astype(np.int64) << b
            if ewid:
                assert not (int(tile_bits) >> epos) & 1
                idx |= ((sbase >> np.uint64(epos)) & np.uint64(1)).astype(np.int64) << b
            seen |= (twid | ewid) << b
        iknib = (np.uint64(T.ik) >> (np.uint64(4) * kk)) & np.uint64(0xf)
        assert np.all((iknib.astype(np.int64) & seen) == 0) and np.all(iknib < np.uint64(1 << T.nbits))
        if not (T.flags & 1):
            assert T.ik == 0
        else:
            assert T.ik != 0
        # every table bit is fed by exactly one of the three parts
        assert (seen | int(np.bitwise_or.reduce(iknib.astype(np.int64)))) == (1 << T.nbits) - 1
        idx |= iknib.astype(np.int64)
        # ... and it is the row's bits at the flipped positions, in ascending order of the index position
        flipped = T.mask_loc if gather else int(deposit(np.array([T.mask_tile], dtype=np.uint64))[0])
        if not gather or desc.n_eff == n_loc:
            pos = [i for i in range(64) if (int(flipped) >> i) & 1]
            if len(pos) == T.nbits:       # (a gather mask of a partitioned operator also flips rank bits: not in mask_loc)
                want = np.zeros(n, dtype=np.int64)
                full_row = (sbase | deposit(tt))
                for b, i in enumerate(pos):
                    want |= ((full_row >> np.uint64(i)) & np.uint64(1)).astype(np.int64) << b
                assert np.array_equal(want, idx)
        coef = quads.tabvals[T.first + idx]
        assert T.z_tile < NT
        par = (_popc(tid & np.uint64(T.z_tile)) + _popc(sbase & np.uint64(T.z_ext))
               + ((np.uint64(T.ksign) >> kk) & np.uint64(1)).astype(np.int64)) & 1
        coef = np.where(par == 1, -coef, coef)
        key = (gather, T.mask_tile, T.mask_loc, T.src)
        assert chain_key is None or key == chain_key, "a chain of table records spans two masks"
        if gather:
            src = xr if T.src else x
            xv = src[(rows ^ np.uint64(T.mask_loc)).astype(np.int64)]
        else:
            assert desc.need_tile
            partner = base | deposit(tt ^ np.uint64(T.mask_tile))
            xv = x[partner.astype(np.int64)]
        acc += coef * xv
        chain_key = None if T.flags & 2 else key
        if q + 1 == desc.tab_loop[1] or q + 1 == desc.tab_loop[2]:
            assert chain_key is None, "the last table record of a mask is not marked"
    y[:] = acc


def run_remote(hm, i, x_recv, y):
    """Apply partner pass i (fed by hm.recvs[i]) to the rank's local y."""
    desc, _ = hm.remote[i]
    y_off = int(desc.sign_base) & ((1 << hm.n_loc) - 1)
    assert x_recv.shape[0] == hm.recvs[i][2] == 1 << desc.n_eff
    run_pass(hm, hm.remote[i], x_recv, y[y_off:y_off + (1 << desc.n_eff)], xr=x_recv)


def multiply(hm, x):
    """Single-rank multiply through all local passes."""
    y = np.zeros(1 << hm.n_loc, dtype=np.complex128)
    for p in hm.local:
        run_pass(hm, p, x, y)
    return y
