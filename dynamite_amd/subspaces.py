"""
Subspaces on which operators and states live: the Python faces of the native
index maps, mirroring ``dynamite.subspaces`` (reference
``src/dynamite/subspaces.py``: Subspace :20-213, Full :214-246, Parity
:248-297, SpinConserve :299-377, Explicit :380-452) for the hot path.
``XParity`` (:532-800) wraps one of them; ``Auto`` (:465-530) is the connected component of a state.

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
        if self._cdesc is None or self._cdesc.vec_swizzle != self.vec_swizzle:
            # (the layout is part of the descriptor: rebuilt if the process state it derives from has changed)
            self._cdesc = self._descriptor()
        return self._cdesc

    def _to_c(self):
        """Mirror of Subspace._to_c (subspaces.py:200-213): what build_mat receives."""
        return {'type': self._enum, 'data': self._c()}

    @property
    def vec_swizzle(self):
        """Layout of this subspace's state vectors in device memory (dnm_subspace.vec_swizzle): Full and Parity
        vectors are XOR-swizzled by ``config.vec_swizzle``; every other subspace keeps index order."""
        from .config import config
        if self._enum not in (FULL, PARITY) or not config.vec_swizzle:
            return 0
        # the swizzle acts on a rank's local index: it needs power-of-two blocks (any other partition of these
        # spaces goes through the index-ordered window path)
        ws = config.world_size
        dim = 1 << (self.L if self._enum == FULL else self.L - 1)
        if ws > 1 and (dim % ws or ((dim // ws) & (dim // ws - 1))):
            return 0
        S = config.vec_swizzle
        from .backend import use_transposed_exchange
        if use_transposed_exchange(ws):
            # the transposed exchange moves pieces of 2^f amplitudes, f = n_local_bits - 1 - p, between layouts, each
            # in ShellMat.TR_SUB = 4 contiguous parts: the swizzle field [S, 2S-4) has to end below the parts
            # (backend.transpose_split, ShellMat._mult_transposed_pipelined)
            p = ws.bit_length() - 1
            cap = ((dim.bit_length() - 1) - 2 * p - 1 - 2 + 4) // 2
            if cap < S:
                # shift 15 is the one value measured slower on MI355X (18.4-19.0 ms at 2^30 amplitudes against
                # 17.4-18.0 for 13, 14 and 16; profiles/r02_exp29_transpose.txt, r02_exp30_transpose_probe.txt)
                S = 14 if cap == 15 else cap
            if S < 5:
                return 0
        return S

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
        d.vec_swizzle = self.vec_swizzle
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
        d.vec_swizzle = self.vec_swizzle
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
        d.vec_swizzle = self.vec_swizzle
        return d

    @property
    def vec_swizzle(self):
        """Layout of this subspace's state vectors (dnm_subspace.vec_swizzle): a | w << 8 for the three-field
        internal layout of csrc/sc3.h (large subspaces; partitions give whole blocks of equal top bits to a rank),
        0 for the reference's index order.  On several ranks the internal layout is taken only where its partition
        is usable (``layout_usable``): otherwise the vectors stay in reference order, split like PETSc splits them, and
        the window kernels run."""
        from .config import config
        lay = config.sc_layout
        if not lay or self.L is None:
            return 0
        a, w = lay
        if self.L - a - w < 1 or math.comb(self.L, self.k) < config.sc_layout_min_dim:
            return 0
        if not self.layout_usable(a, w, config.world_size):
            return 0
        return a | (w << 8)

    # a rank of a partitioned internal layout owns whole blocks of equal top bits T (2^(L - a - w) of them, of very
    # different sizes): with few blocks per rank some ranks get nothing or twice the mean
    LAYOUT_MAX_IMBALANCE = 1.10

    def layout_usable(self, a, w, nranks):
        """Can ``nranks`` ranks share this subspace in the (a, w) internal layout?  One rank: always.  Several: every
        rank must own rows and the largest share may exceed the mean by at most LAYOUT_MAX_IMBALANCE (L=26, k=13 on 4
        ranks would leave a rank empty and give another twice the mean; L=36, k=18 on 8 ranks is balanced to
        0.24 %).  Decided from the host tables of dnm_vec_layout_partition; the same on every rank."""
        key = (self.L, self.k, a, w, max(1, nranks))
        hit = _LAYOUT_USABLE.get(key)
        if hit is None:
            from . import backend
            d = _lib.Subspace()
            d.type, d.L, d.k = SPIN_CONSERVE, self.L, self.k
            d.ld_nchoosek = self.L + 1
            d.nchoosek = _lib.p64(self._nchoosek)
            d.vec_swizzle = a | (w << 8)
            try:
                rows = [backend.layout_partition(d, max(1, nranks), q)[3] for q in range(max(1, nranks))]
            except _lib.BackendError:
                # (an (a, w) the library has no layout for at this (L, k): more top bits than its block tables hold, or a
                # field wider than the subspace fills -- reference order then, like a subspace too small for the layout)
                rows = [0]
            mean = sum(rows) / float(len(rows))
            hit = min(rows) > 0 and max(rows) <= self.LAYOUT_MAX_IMBALANCE * mean
            _LAYOUT_USABLE[key] = hit
        return hit


_LAYOUT_USABLE = {}


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


class Auto(Explicit):
    """The subspace containing ``state`` that the operator ``H`` cannot leave: breadth-first
    search with H as adjacency matrix (subspaces.py:465-530, bsubspace.pyx:212-261).  An edge
    state -> state ^ mask exists when the terms sharing the mask do not cancel on that state.
    ``sort=True`` orders the states numerically; ``sort=False`` keeps the reversed search
    order (reverse Cuthill-McKee), reproduced here level by level."""

    def __init__(self, H, state, size_guess=None, sort=True):
        from . import states, msc_tools
        H.establish_L()
        self._repr_args = f'H={repr(H)}, state={repr(state)}'
        if size_guess is not None:
            self._repr_args += f', size_guess={size_guess}'
        if not sort:
            self._repr_args += ', sort=False'
        self.state = states.State.str_to_state(state, H.L)
        if size_guess is None:
            size_guess = 2 ** H.L
        H.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        signs, coeffs = H.msc['signs'], H.msc['coeffs']
        found = [np.array([self.state], dtype=dnm_int_t)]
        seen = found[0].copy()
        frontier = found[0]
        total = 1
        while frontier.size:
            cand = []      # candidates in the order the serial search meets them: state-major, mask-minor
            for mi in range(masks.size):
                tot = np.zeros(frontier.size, dtype=np.complex128)
                for t in range(offs[mi], offs[mi + 1]):
                    tot += (1 - 2 * msc_tools.parity(frontier & signs[t])) * coeffs[t]
                cand.append(np.where(tot != 0, frontier ^ masks[mi], -1))
            cand = np.stack(cand, axis=1).reshape(-1) if cand else np.empty(0, dtype=dnm_int_t)
            cand = cand[cand >= 0]
            cand = cand[~np.isin(cand, seen)]
            _, first = np.unique(cand, return_index=True)
            new = cand[np.sort(first)].astype(dnm_int_t)
            total += new.size
            if total > size_guess:
                raise RuntimeError('state_map size too small')
            if new.size:
                found.append(new)
                seen = np.union1d(seen, new)
            frontier = new
        state_map = np.concatenate(found)
        if sort:
            state_map.sort()
        else:
            state_map = state_map[::-1]
        Explicit.__init__(self, state_map, L=H.L)

    def __repr__(self):
        return f'Auto({self._repr_args})'


class XParity(Subspace):
    """Symmetric / antisymmetric sector of the global spin flip prod_i sigma^x_i on top of a
    product-state parent subspace (subspaces.py:532-800).  Basis states are
    (|c> +- |c-bar>)/sqrt(2), represented by whichever of c, c-bar has spin L-1 up (bit L-1
    clear), i.e. by the first half of the parent's basis.  The operator is rewritten by
    ``reduce_msc``; the backend then only halves the dimension (bpetsc_template_2.c:223-230)."""

    _product_state_basis = False

    _SECTORS = {'+': +1, '-': -1, +1: +1, -1: -1}

    def __init__(self, parent=None, sector='+', L=None):
        self._parent = Full() if parent is None else parent
        self._chksum = None
        self._cdesc = None
        if L is not None:
            self._parent.L = L
        self._validate_parent(self._parent)
        try:
            self._sector = self._SECTORS[sector]
        except (KeyError, TypeError):
            raise ValueError('invalid value for sector') from None

    @staticmethod
    def _flip_closed(parent):
        """Reason (text) why ``parent`` cannot carry the global flip, or None.  The closed-form spaces are
        decided from their parameters; a listed space (Explicit, Auto) from its states: the first half of the
        basis must be the representatives (spin L-1 up) and the complement of each must be listed too -- the
        second half then holds exactly those complements, since the states are distinct (subspaces.py:566-617)."""
        if isinstance(parent, Parity):
            return None if parent.L % 2 == 0 else 'Parity is only compatible with XParity when L is even'
        if isinstance(parent, SpinConserve):
            return None if parent.L == 2 * parent.k else \
                'SpinConserve is only compatible with XParity when k=L/2'
        half, odd = divmod(parent.get_dimension(), 2)
        if odd:
            return 'parent subspace must have even dimension'
        everything = (1 << parent.L) - 1
        for first in range(0, half, 1 << 16):
            reps = parent.idx_to_state(np.arange(first, min(first + (1 << 16), half)))
            if (reps >> (parent.L - 1)).any():
                return 'first dim/2 basis states must have spin L-1 up (0 in integer notation)'
            if (parent.state_to_idx(reps ^ everything) < 0).any():
                return 'the complement of every state in subspace (all spins flipped) must also be in subspace'
        return None

    @classmethod
    def _validate_parent(cls, parent):
        if not parent.product_state_basis:
            raise ValueError('parent must be a product state subspace')
        if isinstance(parent, Full):
            return                                   # any L, set or not
        if parent.L is None:
            raise ValueError('L must be set for the parent subspace')
        why = cls._flip_closed(parent)
        if why is not None:
            raise ValueError(why)

    @property
    def parent(self):
        return self._parent

    @property
    def sector(self):
        return self._sector

    @property
    def L(self):
        return self.parent.L

    @L.setter
    def L(self, value):
        self.parent.L = value
        self._cdesc = None

    def reduce_msc(self, msc, check_conserves=False):
        """The operator as it acts inside the sector (subspaces.py:632-674).  A Pauli string with an
        odd number of sigma_z/sigma_y factors anticommutes with the global flip and is dropped; a
        string that flips spin L-1 is multiplied by the flip operator (mask complemented, coefficient
        times the sector) so that representatives map to representatives; like terms are merged."""
        from . import msc_tools
        commuting = msc_tools.parity(msc['signs']) == 0
        out = msc[commuting].copy()
        leaves = (out['masks'] >> (self.L - 1)) != 0
        out['masks'][leaves] ^= (1 << self.L) - 1
        out['coeffs'][leaves] *= self.sector
        out = msc_tools.combine_and_sort(out)
        return (out, bool(commuting.all())) if check_conserves else out

    def convert_state(self, state):
        """A state on this subspace -> its parent, or back (subspaces.py:676-762)."""
        from . import states
        state.assert_initialized()
        flip_mask = (1 << self.L) - 1
        half = self.get_dimension()
        v = state.to_numpy(to_all=True)
        if state.subspace is self:
            rtn = states.State(subspace=self.parent)
            out = np.zeros(2 * half, dtype=np.complex128)
            idxs = np.arange(half, dtype=dnm_int_t)
            to_idxs = self.parent.state_to_idx(flip_mask ^ self.idx_to_state(idxs))
            out[to_idxs] = self.sector * v
            out[:half] = v
        elif state.subspace is self.parent:
            rtn = states.State(subspace=self)
            idxs = np.arange(half, 2 * half, dtype=dnm_int_t)
            to_idxs = self.state_to_idx(flip_mask ^ self.parent.idx_to_state(idxs))
            out = np.zeros(half, dtype=np.complex128)
            out[to_idxs] = self.sector * v[half:]
            out += v[:half]
        else:
            raise ValueError('subspace of input state must be this XParity subspace or its parent')
        out /= np.sqrt(2)
        istart, iend = rtn.vec.getOwnershipRange()
        rtn.vec.set_local_from_numpy(out[istart:iend])
        rtn.set_initialized()
        return rtn

    def __eq__(self, s):
        if s is self:
            return True
        if not isinstance(s, XParity):
            if not isinstance(s, Subspace):
                raise ValueError('Cannot compare Subspace to non-Subspace type')
            return False
        return self.sector == s.sector and self.parent == s.parent

    def __hash__(self):
        return hash(('XParity', self.sector, self.parent))

    def __repr__(self):
        return f'XParity({repr(self.parent)}, sector={self.sector:+d})'

    def get_dimension(self):
        return self.parent.get_dimension() // 2

    def idx_to_state(self, idx):
        # representative states are the first N/2 of the parent
        if np.any(np.asarray(idx) >= self.get_dimension()):
            raise ValueError('index out of bounds for this subspace')
        return self.parent.idx_to_state(idx)

    def state_to_idx(self, state):
        if np.count_nonzero(np.asarray(state) >> (self.L - 1)):
            raise ValueError('invalid state')
        return self.parent.state_to_idx(state)

    def _c(self):
        """The parent's descriptor -- with this subspace's own vector layout where the two differ (a SpinConserve
        parent in the internal layout whose XParity vectors stay in reference order: several ranks, a knob)."""
        d = self.parent._c()
        if int(d.vec_swizzle) == self.vec_swizzle:
            return d
        from . import _lib
        if self._cdesc is None or self._cdesc.vec_swizzle != self.vec_swizzle or self._cdesc._parent_desc is not d:
            c = _lib.Subspace.from_buffer_copy(d)
            c.vec_swizzle = self.vec_swizzle
            c._parent_desc = d            # (the copy points into the parent's tables)
            self._cdesc = c
        return self._cdesc

    def _to_c(self):
        return {'type': self.parent._enum, 'data': self._c()}

    @property
    def vec_swizzle(self):
        # an XParity vector is the first half of the parent's in reference order: Full / Parity parents keep their
        # swizzle (it acts on the local index); of a SpinConserve parent's internal layout it is the first half as well
        # (the blocks whose top bit is clear) -- on one rank, where the bond-graph passes of csrc/sc3g_kernels.hip apply
        # the flip-composed hops; several ranks keep reference order
        if isinstance(self.parent, SpinConserve):
            from .config import config
            code = self.parent.vec_swizzle
            if config.world_size != 1 or not config.sc_xparity_layout or code < 256:
                return 0
            from . import _lib
            a, w = code & 0xff, (code >> 8) & 0xff
            return code if (a, w) in ((14, 10), (6, 4)) else 0
        return self.parent.vec_swizzle
