"""
Child of tests/test_gpu_distributed.py::test_c_abi_alone_between_rank_processes: ONE RANK of a partitioned run that binds
the C ABI with ctypes alone -- no torch, no dynamite_amd/backend.py, no torch.distributed: what a Cython / C / MPI host of
include/dynamite_amd.h does.  The 128-byte communicator id comes through a file (an MPI host
would MPI_Bcast it); the transport is whatever DNM_RCCL_LIB names (on a one-GPU box: tests/fake_rccl).

    python native_ranks_child.py CASES.npz RANK WORLD IDFILE OUT.json

For every case of the file with P == WORLD: this rank's handle, its block of x on the device, two
dnm_mat_mult_partitioned calls (the second on the cached schedule) against the oracle's rows; then on the case
DNM_NATIVE_SOLVER_CASE names: dnm_comm_allreduce / the hooks' reductions over the ranks, and the Krylov drivers ACROSS the
rank processes through dnm_comm_hooks -- dnm_eigsolve(nev=1) against the dense solve the parent stored, dnm_expm_multiply
against scipy's.  Replaces bpetsc_template_2.c:413-504, 787-879 and the MFN / EPS calls of computations.py:89-112, 208-257.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from dynamite_amd import _lib as B            # signatures and structs only
    assert "torch" not in sys.modules and "dynamite_amd.backend" not in sys.modules
    fn_cases, rank, world, idfile, fn_out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    L = C.CDLL(os.path.join(ROOT, "dynamite_amd", "libdynamite_amd.so"))
    for name, (res, args) in B.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.dnm_last_error.restype = C.c_char_p

    def ck(rc):
        if rc != 0:
            raise RuntimeError(L.dnm_last_error().decode())
    ck(L.dnm_set_device(0))
    vp = C.c_void_p
    # the communicator: rank 0 draws the id, the others find it in the file
    ident = (C.c_char * 128)()
    if rank == 0:
        ck(L.dnm_comm_unique_id(ident))
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(ident.raw))
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120, "no communicator id"
            time.sleep(0.01)
        ident = (C.c_char * 128).from_buffer_copy(open(idfile, "rb").read())
    comm = vp()
    ck(L.dnm_comm_create(ident, rank, world, C.byref(comm)))

    def dmalloc(nbytes):
        p = vp()
        ck(L.dnm_malloc(C.byref(p), max(16, int(nbytes))))
        return p

    def upload(arr):
        arr = np.ascontiguousarray(arr)
        p = dmalloc(arr.nbytes)
        ck(L.dnm_memcpy_h2d(p, arr.ctypes.data_as(vp), arr.nbytes, None))
        return p

    def download(p, n):
        out = np.empty(n, dtype=np.complex128)
        ck(L.dnm_memcpy_d2h(out.ctypes.data_as(vp), p, out.nbytes, None))
        ck(L.dnm_stream_synchronize(None))
        return out

    # reductions over real ranks
    v = (C.c_double * 3)(rank + 1.0, -float(rank), 0.5)
    ck(L.dnm_comm_allreduce(comm, v, 3, 0))
    assert list(v) == [world * (world + 1) / 2, -world * (world - 1) / 2, 0.5 * world], list(v)
    v = (C.c_double * 2)(float(rank), -float(rank))
    ck(L.dnm_comm_allreduce(comm, v, 2, 1))
    assert list(v) == [world - 1.0, 0.0]

    cases = np.load(fn_cases, allow_pickle=False)
    names = sorted({k.split("/")[0] for k in cases.files})
    report = {"rank": rank, "cases": {}}
    worst = 0.0
    for name in names:
        g = {k.split("/", 1)[1]: cases[k] for k in cases.files if k.startswith(name + "/")}
        if int(g["P"]) != world:
            continue
        typ, Lsp, k, swz = int(g["type"]), int(g["L"]), int(g["k"]), int(g["swz"])
        exchange, flags = int(g["exchange"]), int(g["flags"])
        nck = np.ascontiguousarray(g["nck"], dtype=np.int64)
        sub = B.Subspace()
        sub.type, sub.L, sub.k, sub.space = typ, Lsp, k, int(g["space"])
        sub.ld_nchoosek = Lsp + 1
        sub.nchoosek = nck.ctypes.data_as(B.i64p)
        sub.vec_swizzle = swz
        masks, offs = np.ascontiguousarray(g["masks"]), np.ascontiguousarray(g["mask_offsets"])
        signs, coeffs = np.ascontiguousarray(g["signs"]), np.ascontiguousarray(g["coeffs"])
        x, want = g["x"], g["y"]
        dim = x.size
        internal = typ == 3 and swz >= 256
        part = B.Partition(rank, world)
        if internal:
            a = [C.c_int64() for _ in range(4)]
            ck(L.dnm_vec_layout_partition(C.byref(sub), world, rank, *[C.byref(t) for t in a]))
            s0, rows, nloc = a[2].value, a[3].value, a[1].value
        else:
            qn, rem = divmod(dim, world)
            s0, rows = rank * qn + min(rank, rem), qn + (1 if rank < rem else 0)
            nloc = rows

        def to_device(block):
            nat = upload(block)
            if internal:
                dst = dmalloc(16 * nloc)
                ck(L.dnm_vec_layout_copy(C.byref(sub), C.byref(part), dst, nat, 1, None))
                return dst
            if swz:
                dst = dmalloc(16 * nloc)
                ck(L.dnm_vec_swizzle_copy(dst, nat, nloc, swz, None))
                return dst
            return nat

        def to_host(p):
            if internal:
                nat = dmalloc(16 * rows)
                ck(L.dnm_vec_layout_copy(C.byref(sub), C.byref(part), nat, p, 0, None))
                return download(nat, rows)
            if swz:
                nat = dmalloc(16 * nloc)
                ck(L.dnm_vec_swizzle_copy(nat, p, nloc, swz, None))
                return download(nat, rows)
            return download(p, rows)
        h = vp()
        ck(L.dnm_mat_create(masks.size, B.p64(masks), B.p64(offs), B.p64(signs), coeffs.view(np.float64).ctypes.data_as(B.f64p),
                            C.byref(sub), C.byref(sub), 0, flags, C.byref(part), C.byref(h)))
        if exchange:
            chosen = C.c_int()
            ck(L.dnm_mat_set_exchange(h, exchange, C.byref(chosen)))
            assert chosen.value == exchange, (name, chosen.value)
        if masks.size and masks[0] == 0 and typ == 3:
            ck(L.dnm_mat_precompute_diagonal(h, None))
        xd = to_device(x[s0:s0 + rows])
        if rank % 2 == 0:
            ck(L.dnm_comm_prepare(comm, h, None))           # every other rank: lazily, inside the first multiply
        y = upload(np.full(nloc, 5.0 + 1j, dtype=np.complex128))
        for _ in range(2):
            ck(L.dnm_mat_mult_partitioned(h, comm, xd, y, None))
        ck(L.dnm_stream_synchronize(None))
        err = float(np.abs(to_host(y) - want[s0:s0 + rows]).max())
        worst = max(worst, err / max(1.0, float(np.abs(want).max())))
        report["cases"][name] = {"max_err": err, "rows": rows}

        if name == os.environ.get("DNM_NATIVE_SOLVER_CASE", "sc3_P3"):
            hooks = B.Hooks()
            ck(L.dnm_comm_hooks(comm, h, None, C.byref(hooks)))
            vals = (C.c_double * 1)(rank + 1.0)
            ck(hooks.allreduce_sum(hooks.ctx, vals, 1))
            assert vals[0] == world * (world + 1) / 2
            # Lanczos across the rank processes
            stats = B.SolverStats()
            ev = np.zeros(1)
            ck(L.dnm_eigsolve(h, nloc, 1, B.WHICH["lowest"], 1e-10, 0, 0, 0, C.byref(hooks), 1, B.pf64(ev), None,
                              C.byref(stats), None))
            e0 = float(g["E0"])
            report["eigsolve"] = {"E0": float(ev[0]), "dense": e0, "matvecs": int(stats.matvecs), "residual": float(stats.err_est)}
            assert abs(ev[0] - e0) < 1e-8 * max(1.0, abs(e0)) and stats.nconv >= 1, report["eigsolve"]
            # exp(-i 0.3 H) x across the rank processes
            z = dmalloc(16 * nloc)
            ck(L.dnm_expm_multiply(h, xd, z, nloc, 0.0, -0.3, 1e-10, 0, 0, 1 << 30, C.byref(hooks), C.byref(stats), None))
            wz = g["z"]
            dz = float(np.abs(to_host(z) - wz[s0:s0 + rows]).max())
            report["expm"] = {"max_err": dz, "matvecs": int(stats.matvecs)}
            assert dz < 1e-8, report["expm"]
        ck(L.dnm_comm_forget(comm, h))
        ck(L.dnm_mat_destroy(h))
    ck(L.dnm_comm_destroy(comm))
    report["worst_relative"] = worst
    json.dump(report, open(fn_out, "w"))
    return 0 if worst < 1e-12 else 1


if __name__ == "__main__":
    sys.exit(main())
