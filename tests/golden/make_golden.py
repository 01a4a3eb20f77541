"""
Generates the golden fixtures in this directory by IMPORTING THE REFERENCE's
Python layer (build container only: /root/reference does not exist on the GPU
box, and nothing at test time reads it).

    python tests/golden/make_golden.py

What comes from the reference:
  * the operator algebra (``dynamite.operators``: sigmax/y/z, index_sum,
    op_sum, op_product, ``Operator.reduce_msc`` = msc_tools.combine_and_sort,
    ``Operator._get_mask_offsets``) -> the exact (masks, mask_offsets, signs,
    coeffs) arrays ``bpetsc.build_mat`` would receive (operators.py:615-619);
  * ``msc_tools.msc_to_numpy`` -- the format-defining matrix builder
    (msc_tools.py:1-5,19-92) -> sparse H, from which y = H x, the diagonal,
    the infinity norm, ``scipy.sparse.linalg.expm_multiply`` results (what
    tests/integration/test_evolve.py:54 compares against) and the lowest
    eigenvalues are derived;
  * ``Operator.serialize`` byte strings.

The reference's compiled pieces cannot be built here (no PETSc/SLEPc), so the
absent third-party module ``slepc4py`` and the Cython modules
``dynamite._backend.{bbuild,bsubspace}`` are replaced by inert placeholders
that only provide ``dnm_int_t = int64`` and names; no reference arithmetic
runs through them.  For subspace cases the index maps handed to
``msc_to_numpy`` are the oracle's (oracle/oracle.py), which are themselves
pinned by the reference's unit-test tables in known_answers.json.
"""
import json
import os
import sys
import types

import numpy as np
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.dont_write_bytecode = True


def import_reference():
    # numpy-2: the reference (numpy 1.x era) calls np.array(..., copy=False)
    orig = np.array

    def array(obj, *a, **k):
        if k.get("copy", True) is False:
            k.pop("copy")
            return np.asarray(obj, *a, **k)
        return orig(obj, *a, **k)
    np.array = array

    sl = types.ModuleType("slepc4py")
    sl.init = lambda *a, **k: None
    sys.modules["slepc4py"] = sl

    bb = types.ModuleType("dynamite._backend.bbuild")
    bb.dnm_int_t = np.int64
    bb.have_gpu_shell = lambda: False
    bb.complex_enabled = lambda: True
    bb.petsc_initialized = lambda: False
    bb.get_build_version = lambda: "0.4.0"
    bb.get_build_branch = lambda: ""
    bb.get_build_commit = lambda: ""
    sys.modules["dynamite._backend.bbuild"] = bb

    bs = types.ModuleType("dynamite._backend.bsubspace")
    bs.dnm_int_t = np.int64

    class SubspaceType:
        FULL, PARITY, EXPLICIT, SPIN_CONSERVE = 0, 1, 2, 3
    bs.SubspaceType = SubspaceType
    for n in ("Full", "Parity", "SpinConserve", "Explicit"):
        for f in ("get_dimension", "idx_to_state", "state_to_idx"):
            setattr(bs, f"{f}_{n}", None)
        setattr(bs, "C" + n, None)
    bs.compute_rcm = None
    sys.modules["dynamite._backend.bsubspace"] = bs

    sys.path.insert(0, os.path.join(REF, "src"))
    import dynamite  # noqa: F401
    return dynamite


dynamite = import_reference()
from dynamite import config, msc_tools                     # noqa: E402
from dynamite.operators import (sigmax, sigmay, sigmaz, index_sum, op_sum,   # noqa: E402
                                op_product, Operator)
from dynamite.extras import majorana                        # noqa: E402

sys.path.insert(0, ROOT)
from oracle import oracle as orc                            # noqa: E402


# ---------------------------------------------------------------- hamiltonians

def set_L(L):
    config._L = L


def h_mbl(L):
    """benchmarking/benchmark.py:131-137 ('MBL' = random-field Heisenberg)."""
    from random import seed, uniform
    set_L(L)
    H = index_sum(op_sum(0.25 * s(0) * s(1) for s in (sigmax, sigmay, sigmaz)))
    seed(0)
    for i in range(L):
        H += uniform(-3, 3) * 0.5 * sigmaz(i)
    return H


def h_heisenberg(L):
    """benchmarking/benchmark.py:168-169."""
    set_L(L)
    return index_sum(op_sum(0.25 * s(0) * s(1) for s in (sigmax, sigmay, sigmaz)))


def h_xxz(L, delta=0.5):
    """BASELINE.md section 3 (no reference definition): open chain,
    0.25(XX+YY) + 0.25*delta*ZZ."""
    set_L(L)
    return index_sum(0.25 * sigmax(0) * sigmax(1) + 0.25 * sigmay(0) * sigmay(1)
                     + 0.25 * delta * sigmaz(0) * sigmaz(1))


def h_ising(L):
    """tests/integration/hamiltonians.py:25-31."""
    set_L(L)
    H = index_sum(sigmaz(0) * sigmaz(1), size=L)
    H += 0.5 * index_sum(sigmax(), size=L)
    return H


def h_long_range(L):
    """tests/integration/hamiltonians.py:33-52."""
    set_L(L)
    alpha = 1.13
    H = index_sum(sigmax(0) * sigmax(1), size=L)
    H += op_sum(index_sum(1 / (i ** alpha) * sigmaz(0) * sigmaz(i), size=L)
                for i in range(1, L))
    H += index_sum(0.5 * sigmax(), L)
    H += index_sum(0.3 * sigmay(), L)
    H += index_sum(0.1 * sigmaz(), L)
    return H


def h_localized(L):
    """tests/integration/hamiltonians.py:54-61."""
    set_L(L)
    np.random.seed(0)
    H = index_sum(op_sum(s(0) * s(1) for s in (sigmax, sigmay, sigmaz)), size=L)
    H += op_sum(np.random.uniform(-1, 1) * sigmaz(i) for i in range(L))
    return H


def h_syk(L):
    """tests/integration/hamiltonians.py:63-82."""
    from itertools import combinations
    set_L(L)
    np.random.seed(0)
    maj = [majorana(i) for i in range(L * 2)]

    def gen():
        for idxs in combinations(range(L * 2), 4):
            p = op_product(maj[i] for i in idxs)
            p.scale(np.random.uniform(-1, 1))
            yield p
    return op_sum(gen())


def h_xsum(L):
    """tests/integration/test_eigsolve.py:95-123: sum of sigma_x, spectrum -L+2i."""
    set_L(L)
    return index_sum(sigmax(), size=L)


# ---------------------------------------------------------------- helpers

def random_state(dim, seed=0):
    """states.py:292-316 semantics, single rank."""
    R = np.random.RandomState()
    R.seed(seed % 2 ** 32)
    x = R.standard_normal(dim) + 1j * R.standard_normal(dim)
    return x / np.linalg.norm(x)


def marshal(H):
    H.reduce_msc()
    msc = H.msc.copy()
    masks, offs = H._get_mask_offsets(msc)
    return msc, np.ascontiguousarray(masks), np.ascontiguousarray(offs)


def case_full(name, H, L, ts=(), nev=0, out=None):
    msc, masks, offs = marshal(H)
    dim = 1 << L
    A = msc_tools.msc_to_numpy(msc, (dim, dim)).tocsr()
    x = random_state(dim, 0)
    d = dict(L=L, masks=masks, mask_offsets=offs,
             signs=np.ascontiguousarray(msc["signs"]),
             coeffs=np.ascontiguousarray(msc["coeffs"]),
             x=x, y=A @ x,
             infnorm=float(abs(A).sum(axis=1).max()),
             diag=np.asarray(A.diagonal()),
             hermitian=bool(msc_tools.is_hermitian(msc)),
             serialized=np.frombuffer(msc_tools.serialize(msc), dtype=np.uint8))
    for t in ts:
        key = ("%g%+gj" % (t.real, t.imag)) if isinstance(t, complex) else "%g" % t
        d["expm_t=" + key] = spla.expm_multiply(-1j * t * A.tocsc(), x)
    if nev:
        dense = A.toarray()
        ev = np.linalg.eigvalsh(dense)
        d["evals_lowest"] = ev[:nev]
        d["evals_highest"] = ev[::-1][:nev]
    out[name] = d
    return d


def case_sub(name, H, L, left, right, out):
    """Subspace matrix through the reference builder with pinned index maps."""
    msc, masks, offs = marshal(H)
    M, N = left.dim, right.dim
    A = msc_tools.msc_to_numpy(
        msc, (M, N),
        idx_to_state=lambda r: int(left.i2s(r)[0]),
        state_to_idx=lambda s: right.s2i(s)).tocsr()
    x = random_state(N, 0)
    d = dict(L=L, masks=masks, mask_offsets=offs,
             signs=np.ascontiguousarray(msc["signs"]),
             coeffs=np.ascontiguousarray(msc["coeffs"]),
             x=x, y=A @ x,
             infnorm=float(abs(A).sum(axis=1).max()) if A.nnz else 0.0)
    if M == N:
        d["diag"] = np.asarray(A.diagonal())
    out[name] = d
    return d


def case_xparity(name, H, L, parent_ref, parent_orc, sector, out):
    """XParity(parent, sector): the reference's own ``XParity.reduce_msc``
    (subspaces.py:632-674) rewrites the operator; the matrix is then the reference
    builder's on the first half of the parent's basis (bpetsc_template_2.c:223-230)."""
    from dynamite.subspaces import XParity
    H.reduce_msc()
    msc_in = H.msc.copy()
    sub = XParity(parent_ref, sector=sector)
    red, conserved = sub.reduce_msc(msc_in, check_conserves=True)
    masks, offs = H._get_mask_offsets(red)
    dim = parent_orc.dim // 2
    A = msc_tools.msc_to_numpy(
        red, (dim, dim),
        idx_to_state=lambda r: int(parent_orc.i2s(r)[0]),
        state_to_idx=lambda st: parent_orc.s2i(st)).tocsr()
    x = random_state(dim, 0)
    out[name] = dict(L=L, sector=sector, conserved=bool(conserved),
                     in_masks=np.ascontiguousarray(msc_in["masks"]), in_signs=np.ascontiguousarray(msc_in["signs"]),
                     in_coeffs=np.ascontiguousarray(msc_in["coeffs"]),
                     masks=np.ascontiguousarray(masks), mask_offsets=np.ascontiguousarray(offs),
                     signs=np.ascontiguousarray(red["signs"]), coeffs=np.ascontiguousarray(red["coeffs"]),
                     x=x, y=A @ x, infnorm=float(abs(A).sum(axis=1).max()), diag=np.asarray(A.diagonal()),
                     evals_lowest=np.linalg.eigvalsh(A.toarray())[:4])


def save(fname, cases):
    flat = {}
    for cname, d in cases.items():
        for k, v in d.items():
            flat[cname + "/" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, fname), **flat)
    print("wrote", fname, "%d arrays" % len(flat))


def main():
    full = {}
    case_full("mbl_L6", h_mbl(6), 6, ts=(1.0,), nev=5, out=full)
    case_full("mbl_L10", h_mbl(10), 10, ts=(1.0, 0.3 - 0.2j), nev=5, out=full)
    # BASELINE.json configs[0]: L=12 random-field Heisenberg, evolve(t=1)
    case_full("mbl_L12", h_mbl(12), 12, ts=(1.0, -1.0j * 0.25, 5.0), nev=5, out=full)
    case_full("heisenberg_L10", h_heisenberg(10), 10, ts=(1.0,), nev=5, out=full)
    case_full("xxz_L10", h_xxz(10), 10, ts=(1.0,), nev=5, out=full)
    case_full("ising_L10", h_ising(10), 10, ts=(1.0,), nev=5, out=full)
    case_full("long_range_L8", h_long_range(8), 8, ts=(1.0,), nev=5, out=full)
    case_full("localized_L10", h_localized(10), 10, ts=(1.0,), nev=5, out=full)
    case_full("syk_L5", h_syk(5), 5, ts=(1.0,), nev=5, out=full)
    case_full("xsum_L8", h_xsum(8), 8, nev=3, out=full)
    save("full_space.npz", full)

    sub = {}
    L = 10
    sc5, sc3 = orc.spin_conserve(L, 5), orc.spin_conserve(L, 3)
    pe, po = orc.parity(L, 0), orc.parity(L, 1)
    fu = orc.full(L)
    case_sub("mbl_L10_sc5", h_mbl(L), L, sc5, sc5, sub)
    case_sub("heisenberg_L10_sc3", h_heisenberg(L), L, sc3, sc3, sub)
    case_sub("localized_L10_sc5", h_localized(L), L, sc5, sc5, sub)
    case_sub("ising_L10_parity_even", h_ising(L), L, pe, pe, sub)
    case_sub("ising_L10_parity_odd", h_ising(L), L, po, po, sub)
    case_sub("long_range_L8_parity_even", h_long_range(8), 8, orc.parity(8, 0), orc.parity(8, 0), sub)
    # projections (left != right), test_multiply.py:135-224
    case_sub("mbl_L10_full_to_sc5", h_mbl(L), L, sc5, fu, sub)
    case_sub("mbl_L10_sc5_to_full", h_mbl(L), L, fu, sc5, sub)
    case_sub("ising_L10_even_to_odd", h_ising(L), L, po, pe, sub)
    rs = np.random.RandomState(7)
    states = np.sort(rs.choice(1 << L, size=200, replace=False)).astype(np.int64)
    ex = orc.explicit(L, states)
    perm = rs.permutation(states)
    exu = orc.explicit(L, perm)
    sub["explicit_states"] = dict(sorted=states, unsorted=perm)
    case_sub("long_range_L10_explicit", h_long_range(L), L, ex, ex, sub)
    case_sub("long_range_L10_explicit_unsorted", h_long_range(L), L, exu, exu, sub)
    save("subspaces.npz", sub)

    from dynamite import subspaces as rsub
    xp = {}
    L = 10
    for sector in (+1, -1):
        tag = "plus" if sector == 1 else "minus"
        case_xparity("heisenberg_L10_full_" + tag, h_heisenberg(L), L, rsub.Full(L=L), orc.full(L), sector, xp)
        case_xparity("ising_L10_full_" + tag, h_ising(L), L, rsub.Full(L=L), orc.full(L), sector, xp)
        case_xparity("heisenberg_L10_sc5_" + tag, h_heisenberg(L), L, rsub.SpinConserve(L, 5),
                     orc.spin_conserve(L, 5), sector, xp)
        case_xparity("ising_L10_parity_even_" + tag, h_ising(L), L, rsub.Parity("even", L=L), orc.parity(L, 0),
                     sector, xp)
        case_xparity("ising_L10_parity_odd_" + tag, h_ising(L), L, rsub.Parity("odd", L=L), orc.parity(L, 1),
                     sector, xp)
    # not conserved: sigma_y / sigma_z fields anticommute with the global flip and are dropped
    case_xparity("long_range_L8_full_plus", h_long_range(8), 8, rsub.Full(L=8), orc.full(8), +1, xp)
    case_xparity("mbl_L10_full_minus", h_mbl(L), L, rsub.Full(L=L), orc.full(L), -1, xp)
    save("xparity.npz", xp)


def kagome():
    """Fixtures of the reference's flagship large-scale example (examples/scripts/kagome): `kagome_edges.json` = the
    edge lists `lattice_library.basis_to_graph` produces for every cluster of `kagome_clusters` (data: the bond
    graphs run_kagome.py builds its Hamiltonian on); `kagome.npz` = the Heisenberg operator of run_kagome.py:12-28 on
    the 12- and 15-site tori, built with the reference's operator algebra, in SpinConserve(N, N // 2) (and its XParity
    sectors for N = 12): arrays, y = H x and the lowest eigenvalues from the reference's matrix builder."""
    sys.path.insert(0, os.path.join(REF, "examples", "scripts", "kagome"))
    import lattice_library as ll
    from dynamite import subspaces as rsub
    edges = {}
    for name, basis in ll.kagome_clusters.items():
        verts, e = ll.basis_to_graph(basis)
        edges[name] = {"n": len(verts), "edges": sorted([int(a), int(b)] for a, b in e)}
    with open(os.path.join(HERE, "kagome_edges.json"), "w") as f:
        json.dump(edges, f, separators=(",", ":"))
    print("wrote kagome_edges.json", {k: (v["n"], len(v["edges"])) for k, v in edges.items()})

    def h_kagome(name):
        e = edges[name]
        set_L(e["n"])
        return op_sum(op_sum(0.25 * s(i) * s(j) for s in (sigmax, sigmay, sigmaz))
                      for i, j in e["edges"])
    out = {}
    for name in ("12", "15"):
        N = edges[name]["n"]
        sc = orc.spin_conserve(N, N // 2)
        d = case_sub("kagome_%s_sc" % name, h_kagome(name), N, sc, sc, out)
        msc, _, _ = marshal(h_kagome(name))
        A = msc_tools.msc_to_numpy(msc, (sc.dim, sc.dim), idx_to_state=lambda r: int(sc.i2s(r)[0]),
                                   state_to_idx=lambda st: sc.s2i(st))
        d["evals_lowest"] = np.linalg.eigvalsh(A.toarray())[:6]
    for sector in (+1, -1):
        case_xparity("kagome_12_sc_xparity_" + ("plus" if sector == 1 else "minus"), h_kagome("12"), 12,
                     rsub.SpinConserve(12, 6), orc.spin_conserve(12, 6), sector, out)
    save("kagome.npz", out)


if __name__ == "__main__":
    if "--kagome" in sys.argv:
        kagome()
    else:
        main()
        kagome()
