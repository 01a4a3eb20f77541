"""
``evolve`` and ``eigsolve``: the Krylov callers of the hot path, with the
signatures, defaults and error behaviour of ``dynamite.computations``
(reference ``src/dynamite/computations.py:10-126, 128-292, 511-534``).  The
solvers themselves are native (``dnm_expm_multiply`` / ``dnm_eigsolve``).
"""
import ctypes as C
import warnings

import numpy as np

from . import _lib
from .backend import _stream, _dist
from .config import config, knob


class ConvergenceError(Exception):
    pass


class MaxIterationsError(ConvergenceError):
    pass


def _hooks(mat, keep):
    """Distributed hooks for the native solvers: the partitioned multiply and
    the small all-reduces run through torch.distributed (RCCL)."""
    d = _dist()
    if d is None:
        return None
    import torch
    from .backend import Vec, RawVec, native_comm, native_transport
    if native_transport() and mat._tr is None:
        # the library's own communicator: multiply and reductions of every solver step stay native (dnm_comm_hooks)
        mat._native = native_comm()
        h = _lib.Hooks()
        _lib.check(_lib.lib().dnm_comm_hooks(mat._native, mat.handle, _stream(), C.byref(h)))
        keep.append(h)
        return h

    def mult(ctx, xp, yp):
        try:
            n = mat.n_local
            # wrap raw device pointers as Vec views without copying
            if mat.real_packed:
                x, y = RawVec(_tensor_from_ptr(xp, n), mat.swz_right), RawVec(_tensor_from_ptr(yp, n), mat.swz_left)
            else:
                x = Vec(mat.N, array=_tensor_from_ptr(xp, n), swz=mat.swz_right, sub_c=mat._keep[1])
                y = Vec(mat.M, array=_tensor_from_ptr(yp, n), swz=mat.swz_left, sub_c=mat._keep[0])
            mat.mult(x, y)
            return 0
        except Exception as e:   # pragma: no cover
            print('mult hook failed:', e)
            return 1

    def red(op):
        def f(ctx, buf, n):
            try:
                arr = np.ctypeslib.as_array(buf, shape=(n,))
                t = torch.tensor(arr, dtype=torch.float64, device=config.device)
                d.all_reduce(t, op=op)
                arr[:] = t.cpu().numpy()
                return 0
            except Exception as e:   # pragma: no cover
                print('allreduce hook failed:', e)
                return 1
        return f

    h = _lib.Hooks()
    h.ctx = None
    h.mult = _lib.MULT_FN(mult)
    h.allreduce_sum = _lib.REDUCE_FN(red(d.ReduceOp.SUM))
    h.allreduce_max = _lib.REDUCE_FN(red(d.ReduceOp.MAX))
    keep.append(h)
    return h


def _min_over_ranks(*vals):
    """The smallest of each value over the ranks (decisions taken from free device memory must agree)."""
    d = _dist()
    if d is None:
        return vals if len(vals) > 1 else vals[0]
    import torch
    t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=config.device)
    d.all_reduce(t, op=d.ReduceOp.MIN)
    out = tuple(int(v) for v in t.tolist())
    return out if len(out) > 1 else out[0]


def _tensor_from_ptr(ptr, n):
    """torch complex128 tensor aliasing n amplitudes at a raw device pointer."""
    import torch

    class _W:
        pass
    w = _W()
    w.__cuda_array_interface__ = {'shape': (2 * n,), 'typestr': '<f8', 'data': (int(ptr), False),
                                  'version': 2}
    return torch.as_tensor(w, device=config.device).view(torch.complex128)


def _adopt(mat, state, result):
    """The vectors of ``state`` and ``result`` in the layout ``mat`` works in: where it differs from theirs (a
    relabelled SpinConserve layout, backend._relabelled) the state is converted once and keeps the new layout, the
    result gets a fresh vector in it.  Returns (input Vec, result Vec)."""
    state._vec = mat.vec_in(state.vec)
    if result is not state:
        result._vec = mat.vec_out(result.vec)
    return state.vec, result.vec


def _solver_mat(H, subspace, real):
    """The operator on the partition made for the exchange (Operator.get_solver_mat) if every rank has one, else None."""
    if not config.sc_solver_partition or not H.solver_partition_applies(subspace):
        return None         # (the same answer on every rank: no collective)
    sm = H.get_solver_mat(subspace, real)
    return sm if _min_over_ranks(0 if sm is None else 1) == 1 else None


class _OnSolverPartition:
    """x and y of a solve on the partition made for the exchange: the input state's share moves there block by block
    (backend.reorder_blocks: one vector's worth of traffic, where every multiply of the solve saves about that much --
    SpinConserve(36,18) on 8 ranks: 42 -> 14 GiB on the busiest rank), the result moves back into the State."""

    def __init__(self, sm, xin, yout):
        import torch
        from . import backend
        self.sm, self.yout = sm, yout
        self.x = backend.reorder_blocks(xin.array, xin.sub_c, sm._keep[1])
        self.y = torch.empty_like(self.x)
        self.xptr, self.yptr = C.c_void_p(self.x.data_ptr()), C.c_void_p(self.y.data_ptr())

    @staticmethod
    def applies(sm, xin, yout):
        return (sm is not None and xin.internal and yout.internal and not xin.half and not yout.half
                and xin.perm is None and yout.perm is None
                and (xin.swz & 0xffff) == (yout.swz & 0xffff) == (sm.swz_right & 0xffff))

    def finish(self):
        from . import backend
        self.x = None
        self.yout.array.copy_(backend.reorder_blocks(self.y, self.sm._keep[1], self.yout.sub_c))
        self.y = None


def evolve(H, state, t, result=None, tol=None, ncv=None, algo=None, max_its=None):
    r"""result = exp(-i H t) state   (computations.py:10-126)."""
    state.assert_initialized()
    config._initialize()
    H.establish_L()
    if not H.has_subspace(state.subspace, state.subspace):
        raise ValueError('Hamiltonian and state are defined on different subspaces.')
    if result is None:
        from .states import State
        result = State(L=H.L, subspace=state.subspace)
    elif state.subspace != result.subspace:
        raise ValueError('input and result states are on different subspaces.')
    if algo not in (None, 'expokit', 'krylov', 'chebyshev'):
        raise ValueError("algo must be 'expokit', 'krylov' or 'chebyshev'")
    if t == 0.0:
        state.copy(result)
        return result
    if algo == 'chebyshev':
        return _evolve_chebyshev(H, state, t, result, tol)

    scale = -1j * complex(t)
    mat = H.get_mat(subspaces=(state.subspace, state.subspace))
    keep = []
    hooks = _hooks(mat, keep)
    stats = _lib.SolverStats()
    import torch
    # (an operator on a bond graph works in a relabelled layout of its own: the states adopt it, as in Operator.dot)
    xin, yout = _adopt(mat, state, result)
    mat.check_layout(xin, yout)
    # several ranks, SpinConserve in the internal layout: the iteration runs on the partition made for the exchange
    sm = _solver_mat(H, state.subspace, False)
    side = None
    if _OnSolverPartition.applies(sm, xin, yout):
        side = _OnSolverPartition(sm, xin, yout)
        mat = sm
        hooks = _hooks(mat, keep)
        xptr, yptr = side.xptr, side.yptr
    else:
        xptr, yptr = xin.ptr, yout.ptr
    evolve.last_mat = mat
    mat.prepare_exchange(state.vec.array)
    free, _ = torch.cuda.mem_get_info()
    if free < 34 * 16 * mat.n_local:      # the default basis would not fit: hand torch's cached blocks back first
        torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info()
    # a Krylov basis of fewer than ~10 vectors needs many restarts; for real times the Chebyshev recurrence does
    # the same job with four work vectors
    cached = C.c_size_t()
    _lib.check(_lib.lib().dnm_workspace_bytes(C.byref(cached)))
    fit = int((free * 0.9 + cached.value) // (16 * mat.n_local)) - 2
    fit, free = _min_over_ranks(fit, free)        # Krylov or Chebyshev, and the work limit: the same on every rank
    if ncv is None and fit < 10:
        if complex(t).imag == 0.0 and fit >= 2:
            warnings.warn('evolve: only %d Krylov vectors of %.1f GiB fit in device memory; using '
                          "algo='chebyshev' (4 work vectors)" % (max(fit, 0), 16 * mat.n_local / 2 ** 30),
                          stacklevel=2)
            side = None
            return _evolve_chebyshev(H, state, t, result, tol)
        if fit < 3:
            raise RuntimeError('not enough device memory for a Krylov basis: %d vectors of %.1f GiB fit'
                               % (max(fit, 0), 16 * mat.n_local / 2 ** 30))
    _lib.check(_lib.lib().dnm_expm_multiply(
        mat.handle, xptr, yptr, mat.n_local, scale.real, scale.imag,
        0.0 if tol is None else float(tol), 0 if ncv is None else int(ncv),
        # an explicit algo='krylov' / 'expokit' keeps the Krylov scheme to the end (the driver hands the rest of a
        # real-time interval to the Chebyshev expansion only under all-default parameters)
        (100 if algo is not None else 0) if max_its is None else int(max_its), int(free * 0.9),
        C.byref(hooks) if hooks is not None else None, C.byref(stats), _stream()))
    evolve.last_stats = {'reason': stats.reason, 'its': stats.its, 'matvecs': stats.matvecs,
                         'err_est': stats.err_est}
    # converged-reason mapping of computations.py:114-122
    if stats.reason == _lib.DIVERGED_ITS:
        raise MaxIterationsError('solver reached maximum number of iterations without '
                                 'converging. perhaps try increasing the max iterations with '
                                 'the max_its argument.')
    elif stats.reason == _lib.DIVERGED_BREAKDOWN:
        raise ConvergenceError('solver failed to converge with MFN_DIVERGED_BREAKDOWN.')
    elif stats.reason <= 0:
        raise ConvergenceError('solver failed to converge.')
    if side is not None:
        side.finish()
    result.set_initialized()
    return result


evolve.last_stats = None
evolve.last_mat = None            # the handle the last call multiplied with


def _evolve_chebyshev(H, state, t, result, tol):
    """``algo='chebyshev'`` (not in the reference): the Chebyshev expansion of exp(-iHt) for real t -- one
    multiply and two thirds of a vector sweep per term, four work vectors, no Krylov basis.  Needs about
    ||H||_inf t + 6 (||H||_inf t)^(1/3) + 10 multiplies."""
    if complex(t).imag != 0.0:
        raise ValueError("algo='chebyshev' needs a real time t (use the default Krylov algorithm otherwise)")
    mat = H.get_mat(subspaces=(state.subspace, state.subspace))
    keep = []
    hooks = _hooks(mat, keep)
    stats = _lib.SolverStats()
    xin, yout = _adopt(mat, state, result)
    mat.check_layout(xin, yout)
    sm = _solver_mat(H, state.subspace, False)
    side = None
    xptr, yptr = xin.ptr, yout.ptr
    if _OnSolverPartition.applies(sm, xin, yout):
        side = _OnSolverPartition(sm, xin, yout)
        mat = sm
        hooks = _hooks(mat, keep)
        xptr, yptr = side.xptr, side.yptr
    evolve.last_mat = mat
    mat.prepare_exchange(state.vec.array)
    _lib.check(_lib.lib().dnm_expm_chebyshev(
        mat.handle, xptr, yptr, mat.n_local, float(complex(t).real),
        0.0 if tol is None else float(tol), C.byref(hooks) if hooks is not None else None, C.byref(stats),
        _stream()))
    evolve.last_stats = {'reason': stats.reason, 'its': stats.its, 'matvecs': stats.matvecs,
                         'err_est': stats.err_est}
    if stats.reason <= 0:
        raise ConvergenceError('solver failed to converge.')
    if side is not None:
        side.finish()
    result.set_initialized()
    return result


def eigsolve(H, getvecs=False, nev=1, which='lowest', target=None, tol=None, subspace=None,
             max_its=None, ncv=None, seed=0):
    """A few extremal eigenpairs (computations.py:128-292)."""
    H.establish_L()
    if subspace is None:
        subspace = H.subspace
    elif not H.has_subspace(subspace):
        raise ValueError('Requested subspace has not been added to operator.')
    config._initialize()
    if target is not None:
        # computations.py:211-220: refused for shell matrices and on GPUs
        raise RuntimeError('Shift-invert ("target") not supported for shell matrices.')
    if which == 'target':
        raise ValueError("Must specify target when setting which='target'")
    if which in ['smallest', 'largest']:
        warnings.warn('values "smallest" and "largest" for eigsolve parameter "which" '
                      'are deprecated, and have been replaced by "lowest" and "highest" respectively.',
                      DeprecationWarning, stacklevel=2)
        which = {'smallest': 'lowest', 'largest': 'highest'}[which]
    if which not in _lib.WHICH:
        raise ValueError(f'invalid value "{which}" for which')

    mat = H.get_mat(subspaces=(subspace, subspace))
    # A real-symmetric operator (Heisenberg, XXZ, Ising, random-field chains: every matrix element real in the
    # product basis) needs no complex arithmetic for its eigenpairs: on one rank, from 2^23 amplitudes on, the solver
    # runs on real vectors stored two amplitudes to a complex128 element (DNM_MAT_REAL_PACKED) -- half the bytes per
    # multiply and per Krylov vector, twice the basis in the same memory -- and the eigenvectors are handed back as
    # the complex states the reference returns.  config.eigs_real_arithmetic (DNM_EIGS_REAL=0 / 1 in tests) forces the choice.
    cmat = mat
    er = knob('DNM_EIGS_REAL')
    want_real = config.eigs_real_arithmetic
    if er:
        want_real = er[:1] == '1'
    if want_real is None:
        # (the smallest block decides: blocks of the partitioned SpinConserve layout differ by up to 10 %, and ranks
        # on either side of the threshold would build handles whose exchanges do not match -- ADVICE r4)
        want_real = _min_over_ranks(mat.n_local) >= (1 << 23)
    if want_real:                   # (Full / Parity on 2^p ranks; SpinConserve in the internal layout on any count)
        pm = H.get_real_packed_mat(subspace)
        # every rank or none: a rank whose build was refused keeps its peers on complex128 as well
        if _min_over_ranks(0 if pm is None else 1) == 1:
            mat = pm
    packed = mat is not cmat
    # The solver's own vectors need not lie in the reference's order: on several ranks a SpinConserve operator in the internal
    # layout is solved on a partition made for the exchange (dnm_subspace.vec_swizzle bits 16-19 = 1, csrc/sc3.h: the T
    # blocks ordered so that contiguous ranges cut ONE bond of a chain instead of log2(ranks) + 1 -- SpinConserve(36,18) on
    # 8 ranks: 14.3 GiB to the busiest rank per multiply instead of 42.4).  The start vector holds the numbers the
    # reference order would (keyed by reference index), so the Krylov space is the same one.
    # Eigenvectors move back block by block at the end (backend.reorder_blocks).
    base = mat
    sm = _solver_mat(H, subspace, packed)
    if sm is not None:
        mat = sm
    eigsolve.last_mat = mat
    keep = []
    hooks = _hooks(mat, keep)
    import torch
    vec_bytes = 16 * mat.n_local
    mat.prepare_exchange(torch.empty(0, dtype=torch.complex128, device=config.device))
    # all converged pairs are returned (computations.py:259-281) -- but not at the price of a second basis:
    # large vectors are limited to the nev requested
    nev_max = max(nev, int(ncv) if ncv else max(2 * nev, nev + 15))
    if getvecs and vec_bytes * nev_max > (1 << 30):
        nev_max = nev
    # one extremal pair of a large operator: the native driver runs Lanczos without a stored basis (four work
    # vectors; DNM_EIGS_BASISFREE=0/1 forces the choice) -- nothing to fit into memory then
    import os
    ncv_native = 0 if ncv is None else int(ncv)
    bf = knob('DNM_EIGS_BASISFREE')
    basis_free = (ncv is None and nev == 1 and mat.N > 64 and
                  (bf[:1] == '1' if bf else _min_over_ranks(mat.n_local) >= (1 << 22)))
    if ncv is None and not basis_free:
        # SLEPc's default max(2 nev, nev + 15) (+1 for the residual vector), reduced to what fits in HBM
        cached = C.c_size_t()
        _lib.check(_lib.lib().dnm_workspace_bytes(C.byref(cached)))
        want = max(2 * nev, nev + 15)

        def fitting():
            # (beside the want + 1 basis vectors the driver may take two work vectors for its Chebyshev filter: a
            # multiply's own vectors, freed by the caller but still in torch's cache, once made a solve at
            # SpinConserve(33,16) run out of memory with "17 fit")
            free, _ = torch.cuda.mem_get_info()
            return int((free + cached.value) // vec_bytes) - 3 - (nev_max if getvecs else 0)
        torch.cuda.empty_cache()            # memory torch holds for reuse counts as free
        fit = _min_over_ranks(fitting())
        if fit < want:
            if fit < nev + 5:
                # no room for a restarted basis worth the name (at least nev + 2 vectors beside the Chebyshev filter's
                # work vectors): the native driver takes the pairs one after the other through the basis-free
                # recurrence on the operator deflated by the pairs found -- four work vectors and the pairs themselves
                # (the caller's buffer when the vectors are wanted)
                need = 4 + (0 if getvecs else nev - 1)
                if fit + 3 < need or mat.N <= max(64, 4 * nev):
                    raise RuntimeError('not enough device memory for eigsolve(nev=%d): %d vectors of %.1f GiB fit, '
                                       'the basis-free solver needs %d' % (nev, max(fit + 3, 0), vec_bytes / 2 ** 30,
                                                                           need))
                fit = max(fit, 0)
            # the default basis, capped at what fits (fit + 1 vectors in all): the native driver keeps its automatic
            # choices -- a Chebyshev filter for several pairs of a large operator -- inside that budget
            ncv_native = -(fit + 1)
    evals = np.zeros(nev_max, dtype=np.float64)
    evec_buf = None
    if getvecs:
        from .backend import device_zeros
        evec_buf = device_zeros(nev_max * mat.n_local, empty=True)
    stats = _lib.SolverStats()
    _lib.check(_lib.lib().dnm_eigsolve(
        mat.handle, mat.n_local, int(nev), _lib.WHICH[which], 0.0 if tol is None else float(tol),
        ncv_native, 0 if max_its is None else int(max_its), int(seed),
        C.byref(hooks) if hooks is not None else None, nev_max, _lib.pf64(evals),
        C.c_void_p(evec_buf.data_ptr()) if evec_buf is not None else None, C.byref(stats), _stream()))
    eigsolve.last_stats = {'reason': stats.reason, 'its': stats.its, 'matvecs': stats.matvecs,
                           'nconv': stats.nconv, 'max_rel_residual': stats.err_est, 'real_arithmetic': packed}
    nconv = stats.nconv
    if stats.reason == _lib.DIVERGED_ITS:
        raise MaxIterationsError('eigensolver reached maximum number of iterations without '
                                 'converging. Try increasing the maximum iterations of the '
                                 'eigensolver via the "max_its" argument to eigsolve() '
                                 f'(current value: {max_its})')
    elif stats.reason == _lib.DIVERGED_BREAKDOWN:
        raise ConvergenceError('eigsolver failed to converge with reason EPS_DIVERGED_BREAKDOWN')
    elif stats.reason <= 0 or nconv < nev:
        raise ConvergenceError('eigsolver failed to converge')

    vals = evals[:nconv].copy()
    if not getvecs:
        return vals
    from .states import State
    from .backend import Vec
    evecs = []
    pieces = [evec_buf[i * mat.n_local:(i + 1) * mat.n_local] for i in range(nconv)]
    if mat is not base:
        # from the solver's partition to the one States live on, one vector at a time (the buffer goes as the last leaves)
        from .backend import reorder_blocks
        moved = []
        for i in range(nconv):
            moved.append(reorder_blocks(pieces[i], mat._keep[1], base._keep[1], per_position=1 if packed else 2))
            pieces[i] = None
        pieces, evec_buf, mat = moved, None, base
    for i in range(nconv):
        v = State(L=H.L, subspace=subspace)
        piece = pieces[i]
        if packed:
            # the real eigenvector as the complex state of the full dimension (imaginary parts zero)
            if mat.swz_right >= 256:      # SpinConserve: one double per position of the PACKED handle's own layout
                # (its descriptor and site relabelling, not the complex handle's: the two builds choose theirs
                # separately, and positions of one are not positions of the other -- ADVICE r5)
                v._vec = Vec(cmat.N, swz=mat.swz_right, sub_c=mat._keep[1])
                _lib.check(_lib.lib().dnm_vec_layout_unpack_real(C.byref(mat._keep[1]), C.byref(v._vec._part),
                                                                 v._vec.ptr, C.c_void_p(piece.data_ptr()), _stream()))
                v.set_initialized()
                evecs.append(v)
                continue
            v._vec = Vec(cmat.N, swz=cmat.swz_right, sub_c=cmat._keep[1])
            if v._vec.internal:
                raise RuntimeError('internal: a packed Full / Parity handle beside a SpinConserve layout')
            else:                         # Full / Parity: two amplitudes per element
                _lib.check(_lib.lib().dnm_vec_unpack_real(v._vec.ptr, C.c_void_p(piece.data_ptr()), mat.n_local,
                                                          mat.swz_right, cmat.swz_right, _stream()))
        else:
            # (views of one buffer: no second copy of the vectors)
            v._vec = Vec(mat.N, array=piece, swz=mat.swz_right, sub_c=mat._keep[1])
        v.set_initialized()
        evecs.append(v)
    return vals, evecs


eigsolve.last_stats = None
eigsolve.last_mat = None          # the handle the last call multiplied with (bench.py: its exchange summary)


def reduced_density_matrix(state, keep):
    """Reduced density matrix of ``state`` on the (sorted) spins ``keep``, everything else
    traced out (computations.py:294-349).  Computed on the GPU; returned as a numpy array on
    process 0, ``[[-1]]`` on the other processes, as the reference does."""
    return _reduced_density_matrix(state, keep, on_device=False)


def _reduced_density_matrix(state, keep, on_device):
    """The argument checks of computations.py:294-349, then the kernel; ``on_device``: a device tensor on process 0
    and None elsewhere instead of the host array / ``[[-1]]``."""
    from . import backend
    state.assert_initialized()
    config._initialize()
    if not state.subspace.product_state_basis:
        raise ValueError('reduced density matrices currently only supported '
                         'for product state basis subspace types.')
    keep = np.array(keep, dtype=np.int64)
    if keep.size == 0:
        return np.array([[1]], dtype=np.complex128)
    for n in range(1, keep.size):
        if keep[n] <= keep[n - 1]:
            raise ValueError('keep array must be strictly increasing')
    if any(idx < 0 for idx in keep):
        raise ValueError('spin index less than zero. keep: %s' % str(keep))
    if any(idx >= state.L for idx in keep):
        raise ValueError('spin index greater than spin chain length minus one. keep: %s' % str(keep))
    return backend.reduced_density_matrix(state.vec, state.subspace._to_c(), keep, on_device=on_device)


# reduced density matrices from this size on are diagonalised where they are computed: the spectrum of a
# 2^k x 2^k matrix by the device's dense Hermitian solver (0.9 s at k = 13, where the copy to the host alone takes
# 0.2 s and host LAPACK minutes); smaller ones go through numpy exactly as the reference does
_DEVICE_EIG_FROM = 256


def _rdm_spectrum(state, keep):
    """Eigenvalues of the reduced density matrix (numpy array, ascending) on process 0, None elsewhere."""
    if (1 << len(keep)) < _DEVICE_EIG_FROM:
        reduced = reduced_density_matrix(state, keep)
        if reduced[0, 0] == -1:
            return None
        return np.linalg.eigvalsh(reduced)
    import torch
    rho = _reduced_density_matrix(state, keep, on_device=True)
    if rho is None:
        return None
    from .subspaces import Parity, SpinConserve
    if type(state.subspace) in (SpinConserve, Parity):
        # a state of fixed magnetisation: rho couples only kept configurations with equal numbers of up spins -- the
        # spectrum is that of its blocks (C(k, n) rows each: at 13 of 26 spins 14 blocks of <= 1716 rows instead of
        # one matrix of 8192); a state of fixed parity: equal parities of that number, two blocks of half the size
        K = rho.shape[0]
        a = torch.arange(K, device=rho.device)
        label = torch.zeros(K, dtype=torch.int64, device=rho.device)
        for b in range(len(keep)):
            label += (a >> b) & 1
        if type(state.subspace) is Parity:
            label &= 1
        out = []
        for n in range(int(label.max()) + 1):
            idx = torch.nonzero(label == n).flatten()
            out.append(torch.linalg.eigvalsh(rho.index_select(0, idx).index_select(1, idx)))
        return np.sort(torch.cat(out).cpu().numpy())
    return torch.linalg.eigvalsh(rho).cpu().numpy()


def _entropy_of_spectrum(w):
    log = np.zeros(w.shape)
    np.log(w, where=w > 0, out=log)
    return -np.sum(w * log)


def entanglement_entropy(state, keep):
    """Bipartite entanglement entropy across the cut keep | rest (computations.py:351-383)."""
    w = _rdm_spectrum(state, keep)
    if w is None:                # everything is computed on process 0
        return -1
    return _entropy_of_spectrum(w)


def dm_entanglement_entropy(dm):
    """Von Neumann entropy of a density matrix (computations.py:385-408)."""
    return _entropy_of_spectrum(np.linalg.eigvalsh(dm))


def renyi_entropy(state, keep, alpha, method='eigsolve'):
    """Renyi entropy of the reduced density matrix (computations.py:410-454)."""
    if method == 'eigsolve' or alpha in (0, 1, 'inf'):
        w = _rdm_spectrum(state, keep)
        if w is None:
            return -1
        return _renyi_of_spectrum(w, alpha)
    reduced = reduced_density_matrix(state, keep)
    if reduced[0, 0] == -1:
        return -1
    return dm_renyi_entropy(reduced, alpha, method)


def _renyi_of_spectrum(eigs, alpha):
    if alpha == 0:
        return np.log(np.sum(eigs > 1E-10))
    if alpha == 1:
        return _entropy_of_spectrum(eigs)
    if alpha == 'inf':
        return -np.log(np.max(eigs))
    return 1 / (1 - alpha) * np.log(np.sum(eigs ** alpha))


def dm_renyi_entropy(dm, alpha, method='eigsolve'):
    """H_alpha = log Tr rho^alpha / (1 - alpha), with the alpha = 0, 1, 'inf' limits
    (computations.py:456-507)."""
    if alpha == 0:
        eigs = np.linalg.eigvalsh(dm)
        return np.log(np.sum(eigs > 1E-10))
    if alpha == 1:
        return dm_entanglement_entropy(dm)
    if alpha == 'inf':
        return -np.log(np.max(np.linalg.eigvalsh(dm)))
    if method == 'matrix_power':
        if alpha == int(alpha):
            trace = np.trace(np.linalg.matrix_power(dm, int(alpha))).real
        else:
            raise TypeError('alpha must be an integer for matrix_power method.')
    elif method == 'eigsolve':
        trace = np.sum(np.linalg.eigvalsh(dm) ** alpha)
    else:
        raise ValueError('Valid methods are "eigsolve" and "matrix_power"')
    return 1 / (1 - alpha) * np.log(trace)


def get_tstep(ncv, nrm, tol=1E-7):
    """First-step size of an Expokit solve (computations.py:511-519)."""
    f = ((ncv + 1) / 2.72) ** (ncv + 1) * np.sqrt(2 * np.pi * (ncv + 1))
    t = ((1 / nrm) * (f * tol) / (4.0 * nrm)) ** (1 / ncv)
    s = 10.0 ** (np.floor(np.log10(t)) - 1)
    return np.ceil(t / s) * s


def estimate_compute_time(t, ncv, nrm, tol=1E-7):
    """Cost estimate in matrix multiplies (computations.py:521-528)."""
    return ncv * np.ceil(t / get_tstep(ncv, nrm, tol))
