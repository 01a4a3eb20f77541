"""
CPU ORACLE -- test infrastructure, NOT product code.

ctypes face of ``libdnm_oracle.so`` (``dnm_oracle.c``: the plain-C restatement
of the reference's CPU MatMult / MatNorm / PrecomputeDiagonal / subspace maps)
plus a numpy restatement of the format-defining builder
``msc_tools.msc_to_numpy`` (reference ``src/dynamite/msc_tools.py:19-92``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / the
reported CPU baseline.  Nothing under ``dynamite_amd/`` imports it.

Parity pinning: ``tests/test_oracle.py`` checks every function here against
``tests/golden/known_answers.json`` (tables transcribed from the reference's
unit tests) and ``tests/golden/*.npz`` (vectors produced by importing the
reference's Python layer, see ``tests/golden/make_golden.py``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DNM_LIB_VARIANT=san: the AddressSanitizer + UBSan build (tools/sanitize.sh; the process must LD_PRELOAD the runtime)
_SAN = os.environ.get("DNM_LIB_VARIANT") == "san"
_SO = os.path.join(_HERE, "libdnm_oracle_san.so" if _SAN else "libdnm_oracle.so")

FULL, PARITY, EXPLICIT, SPIN_CONSERVE = 0, 1, 2, 3   # bsubspace_impl.h:17-23

_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


class _Sub(C.Structure):
    _fields_ = [("type", C.c_int), ("L", C.c_int64), ("space", C.c_int64),
                ("k", C.c_int64), ("ld_nchoosek", C.c_int64), ("nchoosek", _i64p),
                ("dim", C.c_int64), ("state_map", _i64p), ("rmap_indices", _i64p),
                ("rmap_states", _i64p), ("xparity", C.c_int64)]


class _Msc(C.Structure):
    _fields_ = [("nmasks", C.c_int64), ("masks", _i64p), ("mask_offsets", _i64p),
                ("signs", _i64p), ("coeffs", C.c_void_p)]


def build(force=False):
    """Compile the oracle with gcc (generic x86-64 flags: the .so travels)."""
    src = [os.path.join(_HERE, f) for f in ("dnm_oracle.c", "dnm_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", os.path.basename(_SO)])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        sp, mp = C.POINTER(_Sub), C.POINTER(_Msc)
        L.orc_dim.restype = C.c_int64
        L.orc_dim.argtypes = [sp]
        L.orc_i2s_array.argtypes = [C.c_int64, sp, _i64p, _i64p]
        L.orc_s2i_array.argtypes = [C.c_int64, sp, _i64p, _i64p]
        L.orc_next_state.restype = C.c_int64
        L.orc_next_state.argtypes = [C.c_int64, C.c_int64, sp]
        L.orc_s2i_nocheck.restype = C.c_int64
        L.orc_s2i_nocheck.argtypes = [C.c_int64, sp]
        L.orc_real_coeffs.argtypes = [mp, _f64p]
        L.orc_precompute_diagonal.argtypes = [mp, sp, _f64p]
        L.orc_matvec_general.argtypes = [mp, sp, sp, _f64p, C.c_void_p, C.c_void_p]
        L.orc_matvec_fast.argtypes = [mp, sp, _f64p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_matvec.argtypes = [mp, sp, sp, _f64p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_matvec_fast_ranks.argtypes = [mp, sp, _f64p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_infnorm.argtypes = [mp, sp, sp, _f64p]
        L.orc_check_conserves.argtypes = [mp, sp, sp, C.POINTER(C.c_int)]
        L.orc_rdm.argtypes = [sp, C.c_void_p, C.c_int64, _i64p, C.c_void_p]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p64(a):
    return a.ctypes.data_as(_i64p)


class Subspace:
    """Holds the numpy buffers a C ``orc_subspace`` points into."""

    def __init__(self, type_, L, space=0, k=0, states=None):
        self.type, self.L, self.space, self.k = type_, int(L), int(space), int(k)
        self._keep = []
        s = _Sub()
        s.type, s.L, s.space, s.k = type_, self.L, self.space, self.k
        if type_ == SPIN_CONSERVE:
            # subspaces.py:340-352: nchoosek[kk, LL] = C(LL, kk), (k+1) x (L+1)
            from math import comb
            tab = np.array([[comb(LL, kk) for LL in range(self.L + 1)]
                            for kk in range(self.k + 1)], dtype=np.int64)
            self.nchoosek = np.ascontiguousarray(tab)
            s.ld_nchoosek = self.L + 1
            s.nchoosek = _p64(self.nchoosek)
        if type_ == EXPLICIT:
            # subspaces.py:394-418
            sm = np.ascontiguousarray(np.asarray(states, dtype=np.int64))
            self.state_map = sm
            if np.all(sm[:-1] <= sm[1:]):
                self.rmap_indices = None
                self.rmap_states = sm
            else:
                self.rmap_indices = np.ascontiguousarray(np.argsort(sm).astype(np.int64))
                self.rmap_states = np.ascontiguousarray(sm[self.rmap_indices])
            s.dim = sm.size
            s.state_map = _p64(self.state_map)
            s.rmap_states = _p64(self.rmap_states)
            if self.rmap_indices is not None:
                s.rmap_indices = _p64(self.rmap_indices)
        self.c = s

    @property
    def ref(self):
        return C.byref(self.c)

    @property
    def dim(self):
        return int(lib().orc_dim(self.ref))

    def i2s(self, idxs):
        idxs = np.ascontiguousarray(np.atleast_1d(idxs), dtype=np.int64)
        out = np.empty_like(idxs)
        lib().orc_i2s_array(idxs.size, self.ref, _p64(idxs), _p64(out))
        return out

    def s2i(self, states):
        states = np.ascontiguousarray(np.atleast_1d(states), dtype=np.int64)
        out = np.empty_like(states)
        lib().orc_s2i_array(states.size, self.ref, _p64(states), _p64(out))
        return out

    def next_state(self, prev, idx):
        return int(lib().orc_next_state(int(prev), int(idx), self.ref))


def full(L):
    return Subspace(FULL, L)


def parity(L, space):
    return Subspace(PARITY, L, space=space)


def spin_conserve(L, k):
    return Subspace(SPIN_CONSERVE, L, k=k)


def explicit(L, states):
    return Subspace(EXPLICIT, L, states=states)


def xparity(parent):
    """The backend's view of XParity(parent): the parent's maps on the first half
    of its indices (bpetsc_template_2.c:223-230).  The operator must have been
    rewritten with ``xparity_reduce_msc`` first."""
    import copy
    s = copy.copy(parent)
    s.c = _Sub.from_buffer_copy(parent.c)
    s.c.xparity = 1
    return s


def xparity_reduce_msc(terms, L, sector):
    """XParity.reduce_msc (subspaces.py:632-674) on a list of (mask, sign, coeff):
    drop terms that anticommute with prod(sigma_x) (odd sign popcount), complement
    masks that flip spin L-1 (times the sector), then merge equal (mask, sign).
    Returns (terms sorted by (mask, sign), conserved)."""
    out, conserved = {}, True
    for m, g, c in terms:
        if _parity(np.array([g], dtype=np.int64))[0]:
            conserved = False
            continue
        c = complex(c)
        if (m >> (L - 1)) & 1:
            m ^= (1 << L) - 1
            if sector == -1:
                c = -c
        out[(m, g)] = out.get((m, g), 0) + c
    # combine_and_sort (msc_tools.py:225-252) drops terms that cancel to zero
    return [(m, g, c) for (m, g), c in sorted(out.items()) if c != 0], conserved


class Msc:
    """(masks, mask_offsets, signs, coeffs) exactly as operators.py:615-619 passes them."""

    def __init__(self, masks, mask_offsets, signs, coeffs):
        self.masks = np.ascontiguousarray(masks, dtype=np.int64)
        self.mask_offsets = np.ascontiguousarray(mask_offsets, dtype=np.int64)
        self.signs = np.ascontiguousarray(signs, dtype=np.int64)
        self.coeffs = np.ascontiguousarray(coeffs, dtype=np.complex128)
        m = _Msc()
        m.nmasks = self.masks.size
        m.masks, m.mask_offsets, m.signs = _p64(self.masks), _p64(self.mask_offsets), _p64(self.signs)
        m.coeffs = self.coeffs.ctypes.data
        self.c = m

    @property
    def ref(self):
        return C.byref(self.c)

    @property
    def nterms(self):
        return int(self.mask_offsets[-1])

    @classmethod
    def from_terms(cls, terms):
        """terms: structured/tuple list of (mask, sign, coeff), already reduced
        and sorted (msc_tools.combine_and_sort); offsets per operators.py:653-669."""
        masks = np.array([t[0] for t in terms], dtype=np.int64)
        signs = np.array([t[1] for t in terms], dtype=np.int64)
        coeffs = np.array([t[2] for t in terms], dtype=np.complex128)
        assert np.all(np.diff(masks) >= 0)
        um, first = np.unique(masks, return_index=True)
        off = np.concatenate([first, [masks.size]]).astype(np.int64)
        return cls(um, off, signs, coeffs)

    def terms(self):
        out = []
        for i, m in enumerate(self.masks):
            for t in range(self.mask_offsets[i], self.mask_offsets[i + 1]):
                out.append((int(m), int(self.signs[t]), complex(self.coeffs[t])))
        return out


def _dptr(diag):
    if diag is None:
        return None
    assert diag.dtype == np.float64 and diag.flags.c_contiguous
    return diag.ctypes.data_as(_f64p)


def real_coeffs(msc):
    out = np.empty(msc.nterms, dtype=np.float64)
    lib().orc_real_coeffs(msc.ref, out.ctypes.data_as(_f64p))
    return out


def precompute_diagonal(msc, sub):
    d = np.empty(sub.dim, dtype=np.float64)
    rc = lib().orc_precompute_diagonal(msc.ref, sub.ref, d.ctypes.data_as(_f64p))
    return None if rc else d


def _xb(x, n):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    return x, np.empty(n, dtype=np.complex128)


def matvec_general(msc, left, right, x, diag=None):
    x, b = _xb(x, left.dim)
    assert x.size == right.dim
    lib().orc_matvec_general(msc.ref, left.ref, right.ref, _dptr(diag), x.ctypes.data, b.ctypes.data)
    return b


def matvec_fast(msc, sub, x, diag=None, nthreads=1):
    x, b = _xb(x, sub.dim)
    rc = lib().orc_matvec_fast(msc.ref, sub.ref, _dptr(diag), x.ctypes.data, b.ctypes.data, nthreads)
    if rc:
        raise ValueError("fast path not applicable")
    return b


def matvec(msc, left, right, x, diag=None, nthreads=1, out=None):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    b = out if out is not None else np.empty(left.dim, dtype=np.complex128)
    lib().orc_matvec(msc.ref, left.ref, right.ref, _dptr(diag), x.ctypes.data, b.ctypes.data, nthreads)
    return b


def matvec_fast_ranks(msc, sub, x, P, diag=None):
    x, b = _xb(x, sub.dim)
    rc = lib().orc_matvec_fast_ranks(msc.ref, sub.ref, _dptr(diag), x.ctypes.data, b.ctypes.data, P)
    if rc:
        raise ValueError("fast ranks path not applicable")
    return b


def infnorm(msc, left, right):
    v = C.c_double()
    lib().orc_infnorm(msc.ref, left.ref, right.ref, C.byref(v))
    return v.value


def check_conserves(msc, left, right):
    r = C.c_int()
    lib().orc_check_conserves(msc.ref, left.ref, right.ref, C.byref(r))
    return bool(r.value)


def rdm(sub, x, keep):
    """Reduced density matrix on the spins ``keep`` (bpetsc_template_1.c:87-165)."""
    keep = np.ascontiguousarray(keep, dtype=np.int64)
    x = np.ascontiguousarray(x, dtype=np.complex128)
    assert x.size == sub.dim
    k = keep.size
    out = np.zeros((1 << k, 1 << k), dtype=np.complex128)
    if lib().orc_rdm(sub.ref, x.ctypes.data, k, _p64(keep), out.ctypes.data):
        raise ValueError('keep array must be strictly increasing')
    return out


def max_threads():
    return int(lib().orc_max_threads())


def fill_test_vectors(x, y, nthreads=1):
    """x = a cheap non-trivial pattern, y = 0, first-touched by the worker threads (bench.py's cpu_baseline)."""
    assert x.dtype == np.complex128 and y.dtype == np.complex128 and x.size == y.size
    L_ = lib()
    L_.orc_fill_test_vectors.restype = None
    L_.orc_fill_test_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    L_.orc_fill_test_vectors(x.ctypes.data, y.ctypes.data, x.size, nthreads)


# ----------------------------------------------------------------------
# numpy restatement of msc_tools.msc_to_numpy (msc_tools.py:19-92)
# ----------------------------------------------------------------------

def _parity(v):
    v = np.asarray(v, dtype=np.uint64).copy()
    for s in (32, 16, 8, 4, 2, 1):
        v ^= v >> np.uint64(s)
    return (v & np.uint64(1)).astype(np.int64)


def msc_to_dense(terms, dims, idx_to_state=None, state_to_idx=None):
    """Dense matrix of an MSC term list.  Row r: ket = idx_to_state(r); for each
    term (m, s, c): bra = ket ^ m, col = state_to_idx(bra) (skip -1),
    H[r, col] += (-1)^popcount(bra & s) * c   (msc_tools.py:63-80)."""
    masks = np.array([t[0] for t in terms], dtype=np.int64)
    signs = np.array([t[1] for t in terms], dtype=np.int64)
    coeffs = np.array([t[2] for t in terms], dtype=np.complex128)
    H = np.zeros(dims, dtype=np.complex128)
    for r in range(dims[0]):
        ket = r if idx_to_state is None else int(idx_to_state(r))
        bra = masks ^ ket
        col = bra if state_to_idx is None else np.asarray(state_to_idx(bra))
        good = np.nonzero(col != -1)[0]
        sgn = 1 - 2 * _parity(signs[good] & bra[good])
        np.add.at(H[r], col[good], sgn * coeffs[good])
    return H
