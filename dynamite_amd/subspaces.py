"""
Subspaces on which operators and states live: the Python faces of the native
index maps, mirroring ``dynamite.subspaces`` (reference
``src/dynamite/subspaces.py``: Subspace :20-213, Full :214-246, Parity
:248-297, SpinConserve :299-377, Explicit :380-452) for the hot path.
``Auto`` and ``XParity`` are not part of this path yet.

All maps go through the C ABI (``dnm_idx_to_state`` / ``dnm_state_to_idx`` /
``dnm_subspace_dim``), which replaces ``bsubspace.pyx``.
"""
import math
from copy import deepcopy
from zlib import crc32

import numpy as np

from . import _lib
from .config import config as _config

FULL, PARITY, EXPLICIT, SPIN_CONSERVE = 0, 1, 2, 3   # bsubspace_impl.h:17-23
dnm_int_t = np.int64


def _validate_L(x):
    # validate.py:6-18
    try:
        if int(x) != x or x < 0:
            raise ValueError()
    except Exception:
        raise ValueError(f'Value must be a nonnegative integer (got "{x!r}")') from None
    if x > 63:
        raise ValueError('Spin chain lengths greater than 63 not supported.')
    return int(x)


class Subspace:
    """Base subspace class (subspaces.py:20-213)."""

    _enum = None
    _product_state_basis = True

    def __init__(self, L=None):
        self._L = None
        self._chksum = None
        self._cdesc = None
        if L is None:
            L = _config.L
        if L is not None:
            self.L = L

    # -- L handling -----------------------------------------------------
    @property
    def L(self):
        return self._L

    def check_L(self, value):
        return value

    @L.setter
    def L(self, value):
        if self._L is not None and value != self._L:
            raise AttributeError('Cannot change L for a subspace after it is set')
        value = _validate_L(value)
        self._L = self.check_L(value)
        self._cdesc = None

    @property
    def product_state_basis(self):
        return self._product_state_basis

    def copy(self):
        return deepcopy(self)

    # -- comparison -----------------------------------------------------
    def __eq__(self, s):
        if s is self:
            return True
        if not isinstance(s, Subspace):
            raise ValueError('Cannot compare Subspace to non-Subspace type')
        if self.L is None:
            raise ValueError('Cannot evaluate equality of subspaces before setting L')
        if self.get_dimension() != s.get_dimension():
            return False
        return self.get_checksum() == s.get_checksum()

    def identical(self, s):
        return hash(self) == hash(s)

    def get_checksum(self):
        # subspaces.py:87-101
        if self._chksum is None:
            BLOCK = 2 ** 14
            chksum = 0
            dim = self.get_dimension()
            for start in range(0, dim, BLOCK):
                stop = min(start + BLOCK, dim)
                chksum = crc32(self.idx_to_state(np.arange(start, stop)), chksum)
            self._chksum = chksum
        return self._chksum

    # -- native descriptor ------------------------------------------------
    def _descriptor(self):
        raise NotImplementedError

    def _c(self):
        """ctypes dnm_subspace (kept alive together with its numpy buffers)."""
        if self.L is None:
            raise ValueError('L has not been set for this subspace')
        if self._cdesc is None:
            self._cdesc = self._descriptor()
        return self._cdesc

    def _to_c(self):
        """Mirror of Subspace._to_c (subspaces.py:200-213): what build_mat receives."""
        return {'type': self._enum, 'data': self._c()}

    # -- maps ----------------------------------------------------------------
    def get_dimension(self):
        import ctypes as C
        d = C.c_int64()
        _lib.check(_lib.lib().dnm_subspace_dim(C.byref(self._c()), C.byref(d)))
        return d.value

    def _map(self, fn, val):
        import ctypes as C
        single = not hasattr(val, "__len__")
        arr = np.ascontiguousarray(np.asarray(val, dtype=dnm_int_t).reshape((-1,)))
        out = np.empty_like(arr)
        try:
            _lib.check(fn(C.byref(self._c()), arr.size, _lib.p64(arr), _lib.p64(out)))
        except _lib.BackendError as e:
            raise ValueError(str(e)) from None   # bsubspace.pyx raises ValueError for bad indices
        return int(out[0]) if single else out

    def idx_to_state(self, idx):
        """Basis index -> integer whose bits are the spin configuration."""
        return self._map(_lib.lib().dnm_idx_to_state, idx)

    def state_to_idx(self, state):
        """Spin configuration -> basis index, or -1 if not in the subspace."""
        return self._map(_lib.lib().dnm_state_to_idx, state)

    def __getstate__(self):
        d = self.__dict__.copy()
        d['_cdesc'] = None
        return d


class Full(Subspace):
    _enum = FULL

    def __init__(self, L=None):
        super().__init__(L)

    def __eq__(self, s):
        if isinstance(s, Full):
            return s.L == self.L
        return super().__eq__(s)

    def __hash__(self):
        return hash((self._enum, self.L))

    def __repr__(self):
        return f'Full(L={self.L})' if self.L is not None else 'Full()'

    def _descriptor(self):
        d = _lib.Subspace()
        d.type, d.L = FULL, self.L
        return d


class Parity(Subspace):
    """Even (0) or odd (1) number of down spins (subspaces.py:248-297)."""
    _enum = PARITY

    def __init__(self, space, L=None):
        self._space = self._check_space(space)
        super().__init__(L)

    @property
    def space(self):
        return self._space

    @classmethod
    def _check_space(cls, value):
        if value in [0, 'even']:
            return 0
        if value in [1, 'odd']:
            return 1
        raise ValueError('Invalid parity space "' + str(value) + '" '
                         '(valid choices are 0, 1, "even", or "odd")')

    def __hash__(self):
        return hash((self._enum, self.L, self.space))

    def __repr__(self):
        arg = {0: "'even'", 1: "'odd'"}[self.space]
        if self.L is not None:
            arg += f', L={self.L}'
        return f'Parity({arg})'

    def _descriptor(self):
        d = _lib.Subspace()
        d.type, d.L, d.space = PARITY, self.L, self.space
        return d


class SpinConserve(Subspace):
    """Fixed number k of down spins (subspaces.py:299-377)."""
    _enum = SPIN_CONSERVE

    def __init__(self, L, k, spinflip=None):
        if spinflip is not None:
            raise DeprecationWarning('spinflip argument has been deprecated; use the XParity '
                                     'class instead.')
        L = _validate_L(L)
        if not (0 <= k <= L):
            raise ValueError('k must be between 0 and L')
        self._k = int(k)
        self._nchoosek = self._compute_nchoosek(L, self._k)
        super().__init__(L=L)

    @classmethod
    def _compute_nchoosek(cls, L, k):
        # (k+1) x (L+1), [kk, LL] = C(LL, kk)  (subspaces.py:340-352)
        rtn = np.empty((k + 1, L + 1), dtype=dnm_int_t)
        for kk in range(k + 1):
            for LL in range(L + 1):
                rtn[kk, LL] = math.comb(LL, kk)
        return np.ascontiguousarray(rtn)

    @property
    def k(self):
        return self._k

    def __hash__(self):
        return hash((self._enum, self.L, self.k))

    def __repr__(self):
        return f'SpinConserve(L={self.L}, k={self.k})'

    def _descriptor(self):
        d = _lib.Subspace()
        d.type, d.L, d.k = SPIN_CONSERVE, self.L, self.k
        d.ld_nchoosek = self.L + 1
        d.nchoosek = _lib.p64(self._nchoosek)
        return d


class Explicit(Subspace):
    """Explicit list of product states (subspaces.py:380-452)."""
    _enum = EXPLICIT

    def __init__(self, state_list, L=None):
        self.state_map = np.ascontiguousarray(np.asarray(state_list, dtype=dnm_int_t))
        if np.all(self.state_map[:-1] <= self.state_map[1:]):
            self.rmap_indices = None
            self.rmap_states = self.state_map
        else:
            self.rmap_indices = np.ascontiguousarray(np.argsort(self.state_map).astype(dnm_int_t))
            self.rmap_states = np.ascontiguousarray(self.state_map[self.rmap_indices])
        if np.any(self.rmap_states[1:] == self.rmap_states[:-1]):
            raise ValueError('values in state_list must be unique')
        super().__init__(L=L)

    def check_L(self, value):
        if int(self.rmap_states[-1]) >> value:
            raise ValueError('State in subspace has more spins than provided')
        return value

    def __hash__(self):
        return hash((self._enum, self.L, self.get_checksum() if self.L is not None else 0))

    def __repr__(self):
        return f'Explicit(<{self.state_map.size} states>, L={self.L})'

    def _descriptor(self):
        d = _lib.Subspace()
        d.type, d.L, d.dim = EXPLICIT, self.L, self.state_map.size
        d.state_map = _lib.p64(self.state_map)
        d.rmap_states = _lib.p64(self.rmap_states)
        if self.rmap_indices is not None:
            d.rmap_indices = _lib.p64(self.rmap_indices)
        return d
