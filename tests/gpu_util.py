"""Helpers shared by the -m gpu tests (they all go through the C ABI)."""
import ctypes as C

import numpy as np

from dynamite_amd import _lib, backend, msc_tools
from dynamite_amd.config import config
from oracle import oracle as orc


def marshal(H):
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    return masks, offs, np.ascontiguousarray(H.msc['signs']), np.ascontiguousarray(H.msc['coeffs'])


def orc_msc(H):
    return orc.Msc(*marshal(H))


def orc_sub(sub):
    from dynamite_amd import subspaces as S
    if isinstance(sub, S.Full):
        return orc.full(sub.L)
    if isinstance(sub, S.Parity):
        return orc.parity(sub.L, sub.space)
    if isinstance(sub, S.SpinConserve):
        return orc.spin_conserve(sub.L, sub.k)
    if isinstance(sub, S.Explicit):
        return orc.explicit(sub.L, sub.state_map)
    raise TypeError(sub)


def shell(H, left, right=None, flags=0, site_perm=None):
    """ShellMat for (left, right) with explicit flags (bypasses Operator's cache)."""
    right = left if right is None else right
    config._initialize()
    m = marshal(H)
    return backend.build_mat(*m, left._to_c(), right._to_c(), flags=flags, site_perm=site_perm)


def vec_from(arr, swz=0, sub_c=None):
    """Device vector holding ``arr`` (index order) in the layout ``swz`` (dnm_subspace.vec_swizzle; the SpinConserve
    internal layout also needs the subspace descriptor)."""
    v = backend.Vec(arr.size, swz=swz, sub_c=sub_c)
    v.set_local_from_numpy(arr)
    return v


def vec_for(sub):
    """Zero vector of the subspace in the layout its states take."""
    swz = sub.vec_swizzle
    return backend.Vec(sub.get_dimension(), swz=swz, sub_c=sub._c() if swz >= 256 else None)


def partner_slice(xl, off, cnt):
    """What a partner rank receives of the local vector ``xl``: the slice [off, off + cnt) of its device array
    (the exchange of ShellMat.mult ships device memory as it lies)."""
    return backend.Vec(cnt, array=xl.array[off:off + cnt], swz=xl.swz)


def mult_numpy(mat, x):
    xv, yv = mat.createVecs()
    xv.set_local_from_numpy(x)
    yv.set(777.0)          # the multiply must overwrite, not accumulate
    mat.mult(xv, yv)
    return yv.local_numpy()


def rand_state(n, seed=0):
    rs = np.random.RandomState(seed)
    return rs.standard_normal(n) + 1j * rs.standard_normal(n)


_SPECTRA = {}


def dense_spectrum(H, sub=None):
    """Ascending eigenvalues of H on ``sub`` (default: the operator's own subspace) by dense diagonalisation on the
    host, computed once per (operator, subspace) and session: several tests solve the same 4096 x 4096 problem, and
    LAPACK on the host is what they spend their time on."""
    import hashlib
    H.establish_L()
    H.reduce_msc()
    key = (hashlib.sha1(H.msc.tobytes()).hexdigest(), repr(sub))
    if key not in _SPECTRA:
        A = H.to_numpy(sparse=False) if sub is None else H.to_numpy(subspaces=(sub, sub), sparse=False)
        _SPECTRA[key] = np.linalg.eigvalsh(A)
    return _SPECTRA[key].copy()
