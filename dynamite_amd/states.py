"""
``State``: the vector container of the hot path, mirroring the part of
``dynamite.states.State`` (reference ``src/dynamite/states.py``) the path uses:
allocation (:102-123), ``set_product`` (:197-241), ``set_uniform`` (:243-251),
``set_random`` (:272-318), ``to_numpy`` (:403-460), BLAS-1 methods (:703-797).
The data live in HBM as one contiguous complex128 block per rank
(``backend.Vec``).
"""
from os import urandom
from time import time

import io
import pickle

import numpy as np

from . import subspaces
from .backend import Vec
from .config import config


class UninitializedError(RuntimeError):
    pass


class State:
    def __init__(self, L=None, subspace=None, state=None, seed=None):
        if L is None:
            L = config.L
        if subspace is None:
            subspace = config.subspace if config.subspace is not None else subspaces.Full()
        self._subspace = subspace.copy() if subspace.L is None else subspace
        if L is not None:
            self._subspace.L = L
        self._L = self._subspace.L
        self._vec = None
        self._initialized = False
        self.repr_binary = False
        if state is not None:
            if state == 'random':
                self.set_random(seed=seed)
            elif state == 'uniform':
                self.set_uniform()
            else:
                self.set_product(state)

    # ------------------------------------------------------------------
    @property
    def L(self):
        return self._L

    @property
    def subspace(self):
        return self._subspace

    @property
    def vec(self):
        if self._vec is None:
            if self.L is None:
                raise ValueError('must set L first')
            swz = self.subspace.vec_swizzle
            self._vec = Vec(self.subspace.get_dimension(), swz=swz, sub_c=self.subspace._c() if swz >= 256 else None)
        return self._vec

    @property
    def initialized(self):
        return self._initialized

    def set_initialized(self):
        self._initialized = True

    def assert_initialized(self):
        if not self._initialized:
            raise UninitializedError('State vector data has not been set yet.')

    def copy(self, result=None):
        self.assert_initialized()
        if result is None:
            result = State(L=self.L, subspace=self.subspace)
        if result.subspace != self.subspace:
            raise ValueError('subspace of state and result must match')
        self.vec.copy(result.vec)
        result.set_initialized()
        return result

    # ------------------------------------------------------------------
    @classmethod
    def str_to_state(cls, s, L):
        """'DUDU..'/'1010..' (leftmost char = spin 0) or int -> int (states.py:147-195)."""
        if isinstance(s, str):
            if len(s) != L:
                raise ValueError('state string must have length L')
            if not all(c in 'UD' for c in s) and not all(c in '01' for c in s):
                raise ValueError('state string must be made of U/D or 0/1')
            state = 0
            for i, c in enumerate(s):
                if c in 'D1':
                    state |= 1 << i
        else:
            state = int(s)
        if state >> L:
            raise ValueError('state has more spins than L')
        return state

    def set_product(self, s):
        if self.L is None and isinstance(s, str):
            self._subspace.L = len(s)
            self._L = len(s)
        idx = self.subspace.state_to_idx(self.str_to_state(s, self.L))
        if idx == -1:
            raise ValueError('Provided initial state not in requested subspace.')
        v = self.vec
        v.set(0)
        istart, iend = v.getOwnershipRange()
        if istart <= idx < iend:
            v.array[v.positions(int(idx - istart))] = 1
        self.repr_binary = isinstance(s, str) and any(c in '01' for c in s)
        self.set_initialized()

    def set_uniform(self):
        self.vec.set(1 / np.sqrt(self.subspace.get_dimension()))
        self.set_initialized()

    def set_random(self, seed=None, normalize=True, device_rng=None):
        """Normalised random state.  Default for local blocks up to 2^26: the
        reference's stream exactly (states.py:292-316: RandomState((seed+rank) %
        2^32), real part drawn first), generated on the host and uploaded.
        Larger blocks (or device_rng=True) use the device's counter-based
        generator: same distribution, different stream."""
        v = self.vec
        if seed is None:
            try:
                seed = int.from_bytes(urandom(4), 'big', signed=False)
            except NotImplementedError:
                seed = int(time())
        if device_rng is None:
            device_rng = v.rows > (1 << 26)
        if device_rng:
            v.set_random(seed)
        else:
            R = np.random.RandomState()
            R.seed((seed + config.rank) % 2 ** 32)
            n = v.rows
            v.set_local_from_numpy(R.standard_normal(n) + 1j * R.standard_normal(n))
        if normalize:
            v.normalize()
        self.set_initialized()

    def set_all_by_function(self, val_fn, vectorize=False):
        """Set every amplitude to ``val_fn(spin configuration)`` (states.py:320-360)."""
        v = self.vec
        istart, iend = v.getOwnershipRange()
        out = np.empty(iend - istart, dtype=np.complex128)
        block = 1 << 16
        for b0 in range(istart, iend, block):
            b1 = min(iend, b0 + block)
            sts = self.subspace.idx_to_state(np.arange(b0, b1))
            if vectorize:
                out[b0 - istart:b1 - istart] = val_fn(sts)
            else:
                out[b0 - istart:b1 - istart] = [val_fn(int(st)) for st in sts]
        v.set_local_from_numpy(out)
        self.set_initialized()

    def project(self, index, value):
        """Projective measurement of spin ``index`` with outcome ``value``, in place, renormalised
        (states.py:364-402)."""
        import torch
        self.assert_initialized()
        if index < 0 or index >= self.L:
            raise ValueError("spin index out of range")
        if value not in (0, 1):
            raise ValueError("value must be 0 or 1")
        v = self.vec
        istart, iend = v.getOwnershipRange()
        block = 1 << 22
        for b0 in range(istart, iend, block):
            b1 = min(iend, b0 + block)
            if isinstance(self.subspace, subspaces.Full):     # index == configuration: mask built on the device
                idx = torch.arange(b0, b1, device=v.array.device)
                kill = ((idx >> index) & 1) != value
            else:
                sts = self.subspace.idx_to_state(np.arange(b0, b1))
                kill = torch.from_numpy(((sts >> index) & 1) != value).to(v.array.device)
            pos = v.positions(torch.arange(b0 - istart, b1 - istart, device=v.array.device))
            v.array[pos[kill]] = 0
        v.normalize()

    def to_numpy(self, to_all=False):
        self.assert_initialized()
        return self.vec.to_numpy(to_all)

    # ------------------------------------------------------------------ files
    # <fname>.metadata: the pickled subspace; <fname>.vec: PETSc's binary Vec format, which is what
    # ``Vec.view`` on a binary viewer writes (states.py:627-650): big-endian, VEC_FILE_CLASSID
    # (1211214), the length, then the entries as (re, im) doubles.  The two header integers are
    # PetscInt-sized: 64-bit for the usual dynamite build (--with-64-bit-indices), 32-bit otherwise;
    # ``int_size`` selects what is written, reading detects it.  (The layout follows PETSc's
    # documentation of its binary format -- no fixture of the reference pins it.)
    VEC_FILE_CLASSID = 1211214
    _IO_CHUNK = 1 << 22

    def save(self, fname, int_size=64):
        self.assert_initialized()
        if int_size not in (32, 64):
            raise ValueError('int_size must be 32 or 64')
        from .backend import _dist
        d = _dist()
        if config.rank == 0:
            with open(fname + '.metadata', 'wb') as f:
                pickle.dump(self.subspace, f)
            with open(fname + '.vec', 'wb') as f:
                f.write(np.array([self.VEC_FILE_CLASSID, self.vec.size], dtype='>i%d' % (int_size // 8)).tobytes())
        if d is not None:
            d.barrier()
        start, end = self.vec.getOwnershipRange()
        with open(fname + '.vec', 'r+b') as f:
            f.seek(2 * (int_size // 8) + 16 * start)
            for lo in range(0, end - start, self._IO_CHUNK):
                hi = min(end - start, lo + self._IO_CHUNK)
                f.write(self.vec.get_local(lo, hi).cpu().numpy().astype('>c16').tobytes())
        if d is not None:
            d.barrier()

    @classmethod
    def from_file(cls, fname):
        """Load a state saved by ``save`` -- or by dynamite itself (states.py:652-701).  Uses
        pickle: do not load files from untrusted sources."""
        with open(fname + '.metadata', 'rb') as f:
            subspace = _SubspaceUnpickler(f).load()
        subspace = _convert_reference_subspace(subspace)
        with open(fname + '.vec', 'rb') as f:
            head = f.read(16)
            h32, h64 = np.frombuffer(head[:8], dtype='>i4'), np.frombuffer(head, dtype='>i8')
            if len(head) == 16 and h64[0] == cls.VEC_FILE_CLASSID:
                n, off = int(h64[1]), 16
            elif len(head) >= 8 and h32[0] == cls.VEC_FILE_CLASSID:
                n, off = int(h32[1]), 8
            else:
                raise RuntimeError("corrupt data encountered when loading state from file")
            if subspace.get_dimension() != n:
                raise RuntimeError("corrupt data encountered when loading state from file")
            rtn = cls(subspace=subspace)
            start, end = rtn.vec.getOwnershipRange()
            import torch
            for lo in range(0, end - start, cls._IO_CHUNK):
                hi = min(end - start, lo + cls._IO_CHUNK)
                f.seek(off + 16 * (start + lo))
                buf = f.read(16 * (hi - lo))
                if len(buf) != 16 * (hi - lo):
                    raise RuntimeError("corrupt data encountered when loading state from file")
                rtn.vec.set_local(lo, hi, torch.from_numpy(np.frombuffer(buf, dtype='>c16').astype(np.complex128))
                                  .to(rtn.vec.array.device))
        rtn.set_initialized()
        return rtn

    def entanglement_entropy(self, keep):
        """states.py:362 -> computations.entanglement_entropy."""
        from . import computations
        return computations.entanglement_entropy(self, keep)

    # ------------------------------------------------------------------ BLAS-1
    def dot(self, x):
        self.assert_initialized()
        x.assert_initialized()
        return self.vec.dot(x.vec)

    def norm(self):
        self.assert_initialized()
        return self.vec.norm()

    def normalize(self):
        self.assert_initialized()
        self.vec.normalize()

    def scale(self, c):
        self.assert_initialized()
        self.vec.scale(c)

    def axpy(self, alpha, x):
        self.scale_and_sum(alpha, 1, x)

    def scale_and_sum(self, alpha, beta, x):
        self.assert_initialized()
        x.assert_initialized()
        if not self.subspace == x.subspace:
            raise ValueError('subspaces do not match')
        if self.vec is x.vec:
            raise ValueError('x and y cannot be the same State object')
        self.vec.axpby(alpha, beta, x.vec)

    def __imul__(self, c):
        self.scale(c)
        return self

    def __mul__(self, c):
        rtn = self.copy()
        rtn *= c
        return rtn

    __rmul__ = __mul__

    def __itruediv__(self, c):
        self.scale(1 / c)
        return self

    def __iadd__(self, x):
        if isinstance(x, State):
            self.axpy(1.0, x)
        else:
            self.assert_initialized()
            self.vec.shift(x)
        return self

    def __add__(self, x):
        rtn = self.copy()
        rtn += x
        return rtn

    __radd__ = __add__

    def __isub__(self, x):
        if isinstance(x, State):
            self.axpy(-1.0, x)
        else:
            self += -x
        return self

    def __sub__(self, x):
        rtn = self.copy()
        rtn -= x
        return rtn

    def __rsub__(self, x):
        rtn = self.copy()
        rtn.scale(-1)
        rtn += x
        return rtn

    def __len__(self):
        return self.subspace.get_dimension()


class _RefSubspace:
    """Stand-in for a pickled ``dynamite.subspaces`` object: only its attribute dict is kept."""
    _ref_name = None


class _SubspaceUnpickler(pickle.Unpickler):
    """Files written by dynamite pickle ``dynamite.subspaces.<Class>`` instances
    (states.py:643-645); they are read as attribute bags and rebuilt as the classes here."""

    def find_class(self, module, name):
        if module.split('.')[0] == 'dynamite' and module.endswith('subspaces'):
            return type(name, (_RefSubspace,), {'_ref_name': name})
        return super().find_class(module, name)


def _convert_reference_subspace(obj):
    if not isinstance(obj, _RefSubspace):
        return obj
    d, name = obj.__dict__, obj._ref_name
    L = d.get('_L')
    if name == 'Full':
        return subspaces.Full(L=L)
    if name == 'Parity':
        return subspaces.Parity(d['_space'], L=L)
    if name == 'SpinConserve':
        return subspaces.SpinConserve(L, d['_k'])
    if name in ('Explicit', 'Auto'):
        return subspaces.Explicit(np.asarray(d['state_map']), L=L)
    if name == 'XParity':
        return subspaces.XParity(_convert_reference_subspace(d['_parent']), sector=d['_sector'])
    raise RuntimeError('unknown subspace type "%s" in state metadata' % name)
