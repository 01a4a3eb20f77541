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
            self._vec = Vec(self.subspace.get_dimension())
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
            v.array[idx - istart] = 1
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
            device_rng = v.local_size > (1 << 26)
        if device_rng:
            v.set_random(seed)
        else:
            R = np.random.RandomState()
            R.seed((seed + config.rank) % 2 ** 32)
            n = v.local_size
            v.set_local_from_numpy(R.standard_normal(n) + 1j * R.standard_normal(n))
        if normalize:
            v.normalize()
        self.set_initialized()

    def to_numpy(self, to_all=False):
        self.assert_initialized()
        return self.vec.to_numpy(to_all)

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

    def __len__(self):
        return self.subspace.get_dimension()
