"""
``Operator`` and the Pauli-string constructors: the caller side of the hot
path, mirroring the parts of ``dynamite.operators`` (reference
``src/dynamite/operators.py``) that the path needs -- operator algebra on the
MSC representation (:938-1142, 1168-1540), ``build_mat`` / ``get_mat`` with the
same marshalling (:546-669), ``dot`` (:1063-1108), ``infinity_norm`` (:206-224),
and ``evolve`` / ``eigsolve`` bound from ``computations`` (:78-79).  String /
LaTeX representations and the stored-matrix (AIJ) path are out of scope.
"""
import warnings

import numpy as np

from . import msc_tools, backend, _lib
from .computations import evolve, eigsolve
from .config import config
from .msc_tools import msc_dtype
from .states import State
from .subspaces import Full, Subspace, XParity


class Operator:
    def __init__(self, msc=None):
        self._max_spin_idx = None
        self._mats = {}
        self._is_reduced = False
        self._shell = True
        self._precompute_diagonal = True
        self._allow_projection = False
        self._msc = None
        self._L = None
        if msc is not None:
            self.msc = msc
        if config.subspace is not None:
            self._subspaces = [(config.subspace, config.subspace)]
        else:
            self._subspaces = [(Full(), Full())]
        if config.L is not None:
            self.L = config.L

    evolve = evolve
    eigsolve = eigsolve

    def copy(self):
        rtn = Operator()
        rtn.msc = self.msc.copy()
        rtn.is_reduced = self.is_reduced
        rtn.shell = self.shell
        rtn._precompute_diagonal = self._precompute_diagonal
        rtn.allow_projection = self.allow_projection
        if self._subspaces is not None:
            rtn._subspaces = [(l.copy(), r.copy()) for l, r in self._subspaces]
        if self._L is not None:
            rtn.L = self._L
        return rtn

    # ------------------------------------------------------------------ sizes
    @property
    def max_spin_idx(self):
        if self._msc is None:
            return -1                  # empty operator (msc_tools.max_spin_idx convention)
        if self._max_spin_idx is None:
            self._max_spin_idx = msc_tools.max_spin_idx(self.msc)
        return self._max_spin_idx

    @property
    def L(self):
        return self._L

    @L.setter
    def L(self, value):
        if value is not None:
            if int(value) != value or value < 0 or value > 63:
                raise ValueError('invalid L')
            value = int(value)
            if value < self.max_spin_idx + 1:
                raise ValueError('Cannot set L smaller than one plus the largest spin index '
                                 'on which the operator has support (max_spin_idx = %d)' %
                                 self.max_spin_idx)
        self._L = value
        if value is not None:
            for left, right in self._subspaces:
                left.L = value
                right.L = value

    def establish_L(self):
        """If L was never set, fix it to the operator's support (operators.py:137-144)."""
        if self._L is None:
            self.L = self.max_spin_idx + 1

    def get_length(self):
        return self.L if self.L is not None else self.max_spin_idx + 1

    @property
    def dim(self):
        return (self.left_subspace.get_dimension(), self.right_subspace.get_dimension())

    @property
    def nnz(self):
        return msc_tools.nnz(self.msc)

    @property
    def nterms(self):
        self.reduce_msc()
        return self.msc.size

    @property
    def msc_size(self):
        return self.msc.size

    @property
    def density(self):
        return self.nnz / self.dim[1]

    # ------------------------------------------------------------------ flags
    @property
    def shell(self):
        return self._shell

    @shell.setter
    def shell(self, value):
        if not isinstance(value, bool):
            raise ValueError('Shell must be set to True or False.')
        if not value:
            # scripts written for the reference pass shell=False for stored matrices: the result is the
            # same operator, applied matrix-free
            warnings.warn('dynamite_amd is matrix-free: shell=False is accepted and ignored', stacklevel=2)
        self._shell = True

    @property
    def precompute_diagonal(self):
        """Cache the diagonal in HBM (+8 B per amplitude read per multiply; default True as in the
        reference, operators.py:246-271).  It is built only for the kernels that read it: the tiled
        hypercube kernel evaluates the diagonal on the fly."""
        return self._precompute_diagonal

    @precompute_diagonal.setter
    def precompute_diagonal(self, value):
        value = bool(value)
        if value != self._precompute_diagonal:
            self.destroy_mat()
        self._precompute_diagonal = value

    @property
    def allow_projection(self):
        return self._allow_projection

    @allow_projection.setter
    def allow_projection(self, value):
        self._allow_projection = bool(value)

    # ------------------------------------------------------------------ subspaces
    @property
    def left_subspace(self):
        return self._subspaces[-1][0]

    @property
    def right_subspace(self):
        return self._subspaces[-1][1]

    @property
    def subspace(self):
        left, right = self._subspaces[-1]
        if not left.identical(right):
            raise ValueError("Left subspace and right subspace are different. Use "
                             "Operator.left_subspace and Operator.right_subspace to access them.")
        return left

    @subspace.setter
    def subspace(self, value):
        self.add_subspace(value, value)

    def add_subspace(self, left, right=None):
        """Register a (left, right) subspace pair (operators.py:307-349); the most
        recently added pair is the default."""
        if right is None:
            right = left
        for sp in (left, right):
            if not isinstance(sp, Subspace):
                raise ValueError('subspace can only be set to objects of Subspace type')
        if left is not right and (not left.product_state_basis or not right.product_state_basis):
            raise ValueError("subspaces must be the same object if either is not a "
                             "product state basis")
        if self.L is None:
            if left.L is not None:
                self.L = left.L
            elif right.L is not None:
                self.L = right.L
        if self.L is not None:
            for sp in (left, right):
                if sp.L is None:
                    sp.L = self.L
                elif sp.L != self.L:
                    raise ValueError('operator and subspaces must all have same spin chain length')
        if not self.has_subspace(left, right):
            self._subspaces.append((left, right))

    def get_subspace_list(self):
        return list(self._subspaces)

    def has_subspace(self, left, right=None):
        if right is None:
            right = left
        return any(l.identical(left) and r.identical(right) for l, r in self._subspaces)

    # ------------------------------------------------------------------ msc
    @property
    def msc(self):
        return self._msc

    @msc.setter
    def msc(self, value):
        self._msc = np.array(value, dtype=msc_dtype).reshape((-1,)).copy() \
            if not isinstance(value, np.ndarray) else np.asarray(value, dtype=msc_dtype)
        self._max_spin_idx = None
        self.is_reduced = False
        self.destroy_mat()

    def reduce_msc(self):
        """Sort, merge and drop zero terms (operators.py:815-822)."""
        if not self.is_reduced:
            self._msc = msc_tools.combine_and_sort(self._msc)
            self.is_reduced = True

    @property
    def is_reduced(self):
        return self._is_reduced

    @is_reduced.setter
    def is_reduced(self, value):
        self._is_reduced = value

    def get_shifted_msc(self, shift, wrap_idx=None):
        return msc_tools.shift(self.msc, shift, wrap_idx)

    def truncate(self, tol=1e-12):
        self.msc = msc_tools.truncate(self.msc, tol)

    def serialize(self):
        self.reduce_msc()
        return msc_tools.serialize(self.msc)

    @classmethod
    def from_bytes(cls, data):
        return cls(msc=msc_tools.deserialize(data))

    def save(self, filename):
        """Write ``serialize()`` to a file (operators.py:505-521); subspaces are not saved."""
        from .backend import _dist
        if config.rank == 0:
            with open(filename, mode='wb') as f:
                f.write(self.serialize())
        if _dist() is not None:
            _dist().barrier()

    @classmethod
    def load(cls, filename):
        """operators.py:523-542."""
        with open(filename, 'rb') as f:
            return cls.from_bytes(f.read())

    def to_numpy(self, subspaces=None, sparse=True):
        """Explicit matrix on the host (small systems; operators.py:869-907)."""
        if subspaces is None:
            subspaces = (self.left_subspace, self.right_subspace)
        self.establish_L()
        self.reduce_msc()
        msc = self.msc if subspaces[0].product_state_basis else subspaces[0].reduce_msc(self.msc)
        return msc_tools.msc_to_numpy(
            msc, (subspaces[0].get_dimension(), subspaces[1].get_dimension()),
            subspaces[0].idx_to_state, subspaces[1].state_to_idx, sparse=sparse)

    # ------------------------------------------------------------------ native matrix
    def get_mat(self, subspaces=None):
        if subspaces is None:
            subspaces = (self.left_subspace, self.right_subspace)
        key = (hash(subspaces[0]), hash(subspaces[1]))
        if key not in self._mats:
            self.build_mat(subspaces)
        return self._mats[key]

    def build_mat(self, subspaces=None):
        """Same sequence as the reference (operators.py:570-631): reduce and sort
        the MSC, Hermiticity gate, unique masks + offsets, hand four contiguous
        arrays and two subspace descriptors to the backend, optionally cache the
        diagonal."""
        if subspaces is None:
            subspaces = (self.left_subspace, self.right_subspace)
        self.establish_L()
        if not self.has_subspace(*subspaces):
            raise ValueError('Attempted to build matrix for a subspace that has not '
                             'been added to the operator.')
        config._initialize()
        self.reduce_msc()
        # XParity is the only non-product-state basis (operators.py:591-594,612-614)
        msc = self.msc if subspaces[0].product_state_basis else subspaces[0].reduce_msc(self.msc)
        if not self.allow_projection and not self.conserves(*subspaces):
            raise ValueError("Constructing the operator's matrix on this subspace yields a "
                             "projection (e.g. subspace is not conserved by the operator). If this "
                             "behavior is desired, set the Operator.allow_projection parameter to True.")
        if not msc_tools.is_hermitian(msc):
            raise ValueError('Building non-Hermitian matrices currently not supported.')
        masks, mask_offsets = msc_tools.get_mask_offsets(msc)
        mat = backend.build_mat(
            masks=np.ascontiguousarray(masks),
            mask_offsets=np.ascontiguousarray(mask_offsets),
            signs=np.ascontiguousarray(msc['signs']),
            coeffs=np.ascontiguousarray(msc['coeffs']),
            left_subspace=subspaces[0]._to_c(),
            right_subspace=subspaces[1]._to_c(),
            xparity=isinstance(subspaces[0], XParity), shell=self.shell, gpu=True)
        # operators.py:627-629.  The tiled hypercube kernel evaluates the diagonal on the fly (cheaper than
        # 8 B/amplitude of extra traffic), so the cache is only built for the kernels that read it.
        if (self.shell and self.precompute_diagonal and subspaces[0] == subspaces[1]
                and masks.size and masks[0] == 0 and mat.uses_cached_diagonal()):
            backend.precompute_diagonal(mat)
        self._mats[(hash(subspaces[0]), hash(subspaces[1]))] = mat

    def get_real_packed_mat(self, subspace):
        """The operator in real arithmetic on ``subspace`` -- Full or Parity on a power-of-two number of ranks (two
        ranks: partner exchange; four and more: the transposed exchange with a swizzle of its own; with XParity on top:
        partner blocks), or SpinConserve in the internal layout on any rank count -- or None when it has an imaginary matrix element in the product basis
        (or the subspace / size has no such form): a second native handle built with ``DNM_MAT_REAL_PACKED`` that multiplies real vectors -- two
        amplitudes to a complex128 element (Full / Parity) or one double per position of the layout (SpinConserve) --
        half the bytes per multiply and per Krylov vector.  Not in the reference (its PETSc build is complex
        throughout); used inside ``eigsolve`` only, which hands back complex states as the reference does."""
        from .subspaces import Full, Parity, SpinConserve, XParity
        key = ('real_packed', hash(subspace))
        if key in self._mats:
            return self._mats[key]
        mat = None
        # (XParity on top of a SpinConserve subspace in the internal layout: the same passes on the layout's first half)
        xp = isinstance(subspace, XParity) and isinstance(subspace.parent, SpinConserve) and subspace.vec_swizzle >= 256
        sc = xp or (isinstance(subspace, SpinConserve) and subspace.vec_swizzle >= 256)
        ws = config.world_size
        # (XParity on top of Full / Parity: the reduced operator on the tiled kernel, packed like any other)
        xpf = isinstance(subspace, XParity) and isinstance(subspace.parent, (Full, Parity)) and ws & (ws - 1) == 0
        ok = sc or xpf or (isinstance(subspace, (Full, Parity)) and ws & (ws - 1) == 0)
        if ok and self.shell:
            self.establish_L()
            self.reduce_msc()
            msc = self.msc if subspace.product_state_basis else subspace.reduce_msc(self.msc)
            masks, mask_offsets = msc_tools.get_mask_offsets(msc)
            sc_desc = subspace._to_c()
            if not sc and ws >= 4:
                # Full / Parity on four or more ranks: the packed vectors are one index bit shorter, and the transposed
                # exchange wants their swizzle field to end below its sub-pieces -- their own shift (they live inside
                # eigsolve only; dnm_vec_unpack_real converts between the two layouts)
                d = type(sc_desc['data']).from_buffer_copy(sc_desc['data'])
                S = int(d.vec_swizzle)
                bits = (self.L if isinstance(subspace, Full) else self.L - 1) - 1
                p = ws.bit_length() - 1
                cap = (bits - 2 * p - 1 - 2 + 4) // 2
                if S and cap < S:
                    S = 14 if cap == 15 else cap
                d.vec_swizzle = S if S >= 5 else 0
                sc_desc = {'type': sc_desc['type'], 'data': d}
            try:
                mat = backend.build_mat(
                    masks=np.ascontiguousarray(masks), mask_offsets=np.ascontiguousarray(mask_offsets),
                    signs=np.ascontiguousarray(msc['signs']), coeffs=np.ascontiguousarray(msc['coeffs']),
                    left_subspace=sc_desc, right_subspace=sc_desc, xparity=xp or xpf, flags=_lib.MAT_REAL_PACKED)
            except _lib.BackendError as e:
                # the native layer's refusals all name the form: an imaginary matrix element, no chain operator in the
                # SpinConserve layout, a vector too small for the tiled kernel.  Anything else (memory, a bad table) is
                # an error of the build, not a reason to fall back to complex arithmetic silently
                if 'real-packed' not in str(e):
                    raise
                mat = None
            if mat is not None and masks.size and masks[0] == 0 and mat.uses_cached_diagonal():
                backend.precompute_diagonal(mat)        # (more mixed diagonal patterns than the passes evaluate on the fly)
        self._mats[key] = mat
        return mat

    SOLVER_BLOCK_ORDER = 1

    def solver_partition_applies(self, subspace):
        """Could ``get_solver_mat`` give an operator here?  What every rank answers alike without building anything: several
        ranks, a SpinConserve subspace in the internal layout, a shell operator."""
        from .subspaces import SpinConserve
        return bool(config.world_size > 1 and isinstance(subspace, SpinConserve)
                    and 256 <= subspace.vec_swizzle < (1 << 16) and self.shell)

    def get_solver_mat(self, subspace, real):
        """The operator on ``subspace`` for vectors that live INSIDE a solver (no state of the caller's ever meets them):
        SpinConserve in the internal layout on several ranks, its T blocks in the order made for partitions
        (``dnm_subspace.vec_swizzle`` bits 16-19 = 1, csrc/sc3.h) -- a rank's contiguous share is no range of the
        reference order, but its rows read far less from other ranks.  None where that does not apply (one rank, other
        subspaces, operators that need a cached diagonal: its rows are computed in reference order)."""
        from .subspaces import SpinConserve
        key = ('solver', hash(subspace), bool(real))
        if key in self._mats:
            return self._mats[key]
        mat = None
        if self.solver_partition_applies(subspace):
            self.establish_L()
            self.reduce_msc()
            masks, mask_offsets = msc_tools.get_mask_offsets(self.msc)
            desc = subspace._to_c()
            d = type(desc['data']).from_buffer_copy(desc['data'])
            d.vec_swizzle = int(d.vec_swizzle) | (self.SOLVER_BLOCK_ORDER << 16)
            sd = {'type': desc['type'], 'data': d, '_keep': desc}
            ws = config.world_size
            rows = [backend.layout_partition(d, ws, q)[3] for q in range(ws)]
            if min(rows) <= 0 or max(rows) > 1.5 * sum(rows) / ws:
                self._mats[key] = None        # (few, uneven blocks: this order has nothing to share out)
                return None
            try:
                mat = backend.build_mat(
                    masks=np.ascontiguousarray(masks), mask_offsets=np.ascontiguousarray(mask_offsets),
                    signs=np.ascontiguousarray(self.msc['signs']), coeffs=np.ascontiguousarray(self.msc['coeffs']),
                    left_subspace=sd, right_subspace=sd, flags=_lib.MAT_REAL_PACKED if real else 0, site_perm=False)
            except _lib.BackendError as e:
                if 'real-packed' not in str(e):
                    raise
                mat = None
            if mat is not None and (mat.uses_cached_diagonal() or 'internal layout' not in mat.describe()):
                mat.destroy()
                mat = None
        self._mats[key] = mat
        return mat

    def destroy_mat(self, subspaces=None):
        keys = [(hash(subspaces[0]), hash(subspaces[1]))] if subspaces is not None else list(self._mats)
        if subspaces is not None:
            keys.append(('real_packed', hash(subspaces[0])))
            keys += [('solver', hash(subspaces[0]), r) for r in (False, True)]
        for k in keys:
            mat = self._mats.pop(k, None)
            if mat is not None:
                mat.destroy()

    def conserves(self, left, right=None):
        """Does the operator map the right subspace into the left one?
        (operators.py:382-423 -> bpetsc.check_conserves -> CheckConserves,
        bpetsc_template_2.c:990-1056; a full sweep over the columns, on the GPU.)"""
        if right is None:
            right = left
        if not left.product_state_basis or not right.product_state_basis:
            if left is not right:
                raise ValueError('if left or right subspace is not a product '
                                 'state basis, they must be the same object')
        if isinstance(left, Full) and isinstance(right, Full):
            return True
        self.establish_L()
        for sp in (left, right):
            if sp.L is None:
                sp.L = self.L
        self.reduce_msc()
        if not left.product_state_basis:
            msc, conserved = left.reduce_msc(self.msc, check_conserves=True)
            if not conserved:
                return False
        else:
            msc = self.msc
        masks, offs = msc_tools.get_mask_offsets(msc)
        return backend.check_conserves(masks, offs, np.ascontiguousarray(msc['signs']),
                                       np.ascontiguousarray(msc['coeffs']),
                                       left._to_c(), right._to_c(), xparity=isinstance(left, XParity))

    def infinity_norm(self, subspaces=None):
        return self.get_mat(subspaces=subspaces).norm('infinity')

    def create_states(self):
        self.establish_L()
        return (State(L=self.L, subspace=self.left_subspace),
                State(L=self.L, subspace=self.right_subspace))

    def dot(self, x, result=None):
        r"""y = A x  (operators.py:1063-1108)."""
        x.assert_initialized()
        self.establish_L()
        right_subspace = x.subspace
        right_match = [(l, r) for l, r in self.get_subspace_list() if r.identical(right_subspace)]
        if not right_match:
            raise ValueError('No operator subspace found that matches input vector subspace. '
                             'Try adding the subspace with the Operator.add_subspace method.')
        if result is None:
            if len(right_match) != 1:
                raise ValueError('Ambiguous subspace for result vector. Pass a state with the '
                                 'desired subspace as the "result" option to Operator.dot.')
            left_subspace = right_match[0][0]
            result = State(L=left_subspace.L, subspace=left_subspace)
        else:
            left_subspace = result.subspace
        if not any(l.identical(left_subspace) for l, _ in right_match):
            raise ValueError('Subspaces of matrix and result vector do not match.')
        mat = self.get_mat(subspaces=(left_subspace, right_subspace))
        # An operator on a bond graph works in a relabelled layout of its own (backend._relabelled): the states ADOPT
        # it -- x is converted once (it keeps its content; every State operation converts where layouts meet) and
        # the result is created in it, so that repeated products with this operator move nothing through the
        # reference order.
        x._vec = mat.vec_in(x.vec)
        if result is not x:
            result._vec = mat.vec_out(result.vec)
        mat.mult(x.vec, result.vec)
        result.set_initialized()
        return result

    def expectation(self, state, tmp_state=None):
        """<state|A|state> (operators.py:775-796)."""
        if tmp_state is None:
            tmp_state = self.dot(state)
        else:
            self.dot(state, tmp_state)
        return state.dot(tmp_state).real

    # ------------------------------------------------------------------ algebra
    def _check_compatible(self, o):
        if self.L != o.L:
            raise ValueError("Operators to be combined must have the same value of the spin chain "
                             "length L. To set it globally, set config.L")
        if self.allow_projection != o.allow_projection:
            raise ValueError("Operators must have the same value of the 'allow_projection' "
                             "parameter to be combined.")

    def __add__(self, x):
        if not isinstance(x, Operator):
            if x == 0:
                return self.copy()
            x = x * identity()
        self._check_compatible(x)
        rtn = self.copy()
        rtn.msc = msc_tools.msc_sum([self.msc, x.msc])
        return rtn

    def __radd__(self, x):
        return self.__add__(x)

    def __sub__(self, x):
        return self + -x

    def __rsub__(self, x):
        return x + -self

    def __neg__(self):
        return -1 * self

    def __mul__(self, x):
        if isinstance(x, Operator):
            self._check_compatible(x)
            rtn = self.copy()
            rtn.msc = msc_tools.msc_product([self.msc, x.msc])
            return rtn
        if isinstance(x, State):
            return self.dot(x)
        return self._num_mul(x)

    def __rmul__(self, x):
        if isinstance(x, State):
            raise TypeError('Left vector-matrix multiplication not currently supported.')
        return self._num_mul(x)

    def __truediv__(self, x):
        if isinstance(x, Operator):
            raise TypeError('Dividing by Operators not supported.')
        return (1 / x) * self

    def __eq__(self, x):
        if isinstance(x, Operator):
            self.reduce_msc()
            x.reduce_msc()
            return np.array_equal(self.msc, x.msc)
        raise TypeError('Equality not supported for types %s and %s' % (type(self), type(x)))

    __hash__ = None

    def scale(self, x):
        if x == 1:
            return
        try:
            self.msc['coeffs'] *= x
        except (ValueError, TypeError):
            raise TypeError(f'Cannot scale operator by type {type(x)}')
        self.destroy_mat()

    def _num_mul(self, x):
        rtn = self.copy()
        rtn.scale(x)
        return rtn


# ---------------------------------------------------------------------- constructors

def _spin_index(i):
    if int(i) != i or i < 0 or i > 62:
        raise ValueError('invalid spin index')
    return int(i)


def sigmax(i=0):
    r""":math:`\sigma^x_i`  = msc (1<<i, 0, 1)   (operators.py:1426-1440)."""
    i = _spin_index(i)
    return Operator(msc=[(1 << i, 0, 1)])


def sigmay(i=0):
    r""":math:`\sigma^y_i`  = msc (1<<i, 1<<i, 1j)   (operators.py:1442-1456)."""
    i = _spin_index(i)
    return Operator(msc=[(1 << i, 1 << i, 1j)])


def sigmaz(i=0):
    r""":math:`\sigma^z_i`  = msc (0, 1<<i, 1)   (operators.py:1458-1472)."""
    i = _spin_index(i)
    return Operator(msc=[(0, 1 << i, 1)])


def sigma_plus(i=0):
    return sigmax(i) + 1j * sigmay(i)


def sigma_minus(i=0):
    return sigmax(i) - 1j * sigmay(i)


def identity():
    return Operator(msc=[(0, 0, 1)])


def zero():
    return Operator(msc=np.empty(0, dtype=msc_dtype))


def op_sum(terms, nshow=3):
    """Sum of operators (operators.py:1168-1218)."""
    terms = list(terms)
    if not terms:
        return zero()
    rtn = Operator(msc=msc_tools.msc_sum(t.msc for t in terms))
    if terms[0].L is not None:
        rtn.L = terms[0].L
    return rtn


def op_product(terms):
    """Product of operators, in order (operators.py:1220-1264)."""
    terms = list(terms)
    if not terms:
        return identity()
    rtn = Operator(msc=msc_tools.msc_product(t.msc for t in terms))
    if terms[0].L is not None:
        rtn.L = terms[0].L
    return rtn


def index_sum(op, size=None, start=0, boundary='open'):
    """Translate ``op`` along the chain and sum (operators.py:1266-1349)."""
    if size is None:
        if op.L is None:
            raise ValueError('Must specify index_sum size with either the "size" argument '
                             'or by setting Operator.L (possibly through config.L).')
        size = op.L
    if boundary == 'open':
        stop = start + size - op.max_spin_idx
        if stop <= start:
            raise ValueError("requested size %d for sum operator's support smaller than "
                             "summand's support %d; impossible to satisfy" % (size, op.max_spin_idx))
        wrap_idx = None
    elif boundary == 'closed':
        stop = start + size
        wrap_idx = stop
        if start != 0:
            raise ValueError('cannot set start != 0 for closed boundary conditions.')
    else:
        raise ValueError("invalid value for argument 'boundary' (can be 'open' or 'closed')")
    rtn = Operator(msc=msc_tools.msc_sum(op.get_shifted_msc(i, wrap_idx) for i in range(start, stop)))
    if op.L is not None:
        rtn.L = op.L
    return rtn


def index_product(op, size=None, start=0):
    """Translate ``op`` along the chain and multiply (operators.py:1351-1411)."""
    if size is None:
        if op.L is None:
            raise ValueError('Must specify index_product size')
        size = op.L
    if size == 0:
        return identity()
    stop = start + size - op.max_spin_idx
    return op_product(Operator(msc=op.get_shifted_msc(i, None)) for i in range(start, stop))
