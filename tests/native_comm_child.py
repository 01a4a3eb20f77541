"""
Child process of tests/test_gpu_distributed.py::test_native_partitioned_multiply: the partitioned multiply as ONE native
call (dnm_mat_mult_partitioned, dynamite_amd/csrc/comm.cpp) bound with ctypes alone -- no torch, no
dynamite_amd/backend.py: what a Cython binding of include/dynamite_amd.h would do.  Device
memory comes from dnm_malloc, the communicator from dnm_comm_unique_id / dnm_comm_create (RCCL, world size 1), and
dnm_comm_loopback makes this process stand for rank r of P: the peers' blocks of x live in this process and every
message of the exchange is an RCCL send to itself -- the schedule and the transport on a one-GPU box.

Cases come from the parent as an .npz (operator arrays, subspace parameters, the global x and the oracle's y = H x):
  partner  Full space on P = 2 / 8 ranks, swizzled vectors (XOR-partner sub-blocks, bpetsc_template_2.c:787-879);
  window   SpinConserve in reference order on 3 ranks (rows split like PetscSplitOwnership, row-range overlap), Full
           on 3 ranks, SpinConserve in the internal layout on 2 / 3 ranks (two tiled passes split around the exchange).
  transpose  Full / Parity on 2 / 4 / 8 ranks under dnm_mat_set_exchange(DNM_EXCHANGE_TRANSPOSE): the operator split
           inside the handle, the state through one all-to-all, the second part sub-piece by sub-piece (and in one
           piece: DNM_TRANSPOSE_PIPE=0), the returning pieces -- what the PEERS' second parts computed, run by the
           loop-back from their handles -- added on arrival; also a real-packed operator (DNM_MAT_REAL_PACKED).
Then the solvers through dnm_comm_hooks on one of the partitions.  Prints one JSON line; exit code 0 = all equal.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from dynamite_amd import _lib as B            # signatures and structs only: the library is loaded right here
    assert "torch" not in sys.modules and "dynamite_amd.backend" not in sys.modules
    L = C.CDLL(os.path.join(ROOT, "dynamite_amd", "libdynamite_amd.so"))
    for name, (res, args) in B.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.dnm_last_error.restype = C.c_char_p

    def ck(rc):
        if rc != 0:
            raise RuntimeError(L.dnm_last_error().decode())

    ck(L.dnm_set_device(0))
    cases = np.load(sys.argv[1], allow_pickle=False)
    names = sorted({k.split("/")[0] for k in cases.files})
    vp = C.c_void_p

    def dmalloc(nbytes):
        p = vp()
        ck(L.dnm_malloc(C.byref(p), max(16, int(nbytes))))
        return p

    def upload(arr):
        arr = np.ascontiguousarray(arr)
        p = dmalloc(arr.nbytes)
        ck(L.dnm_memcpy_h2d(p, arr.ctypes.data_as(vp), arr.nbytes, None))
        return p

    def download(p, n, dtype=np.complex128):
        out = np.empty(n, dtype=dtype)
        ck(L.dnm_memcpy_d2h(out.ctypes.data_as(vp), p, out.nbytes, None))
        ck(L.dnm_stream_synchronize(None))
        return out

    ident = (C.c_char * 128)()
    ck(L.dnm_comm_unique_id(ident))
    comm = vp()
    ck(L.dnm_comm_create(ident, 0, 1, C.byref(comm)))
    v = (C.c_double * 3)(1.5, -2.0, 7.0)
    ck(L.dnm_comm_allreduce(comm, v, 3, 0))
    ck(L.dnm_comm_allreduce(comm, v, 3, 1))
    assert list(v) == [1.5, -2.0, 7.0]

    report = {"cases": {}}
    worst = 0.0
    for name in names:
        g = {k.split("/", 1)[1]: cases[k] for k in cases.files if k.startswith(name + "/")}
        typ, Lsp, k, P = int(g["type"]), int(g["L"]), int(g["k"]), int(g["P"])
        swz = int(g["swz"])
        exchange, flags, packed = int(g["exchange"]), int(g["flags"]), bool(int(g["flags"]) & B.MAT_REAL_PACKED)
        if "env" in g:
            for kv in str(g["env"]).split():
                os.environ[kv.split("=")[0]] = kv.split("=")[1]
        nck = np.ascontiguousarray(g["nck"], dtype=np.int64)
        sub = B.Subspace()
        sub.type, sub.L, sub.k = typ, Lsp, k
        sub.space = int(g["space"])
        sub.ld_nchoosek = Lsp + 1
        sub.nchoosek = nck.ctypes.data_as(B.i64p)
        sub.vec_swizzle = swz
        masks, offs = np.ascontiguousarray(g["masks"]), np.ascontiguousarray(g["mask_offsets"])
        signs, coeffs = np.ascontiguousarray(g["signs"]), np.ascontiguousarray(g["coeffs"])
        x, want = g["x"], g["y"]
        if packed:             # real vectors, two amplitudes to an element
            assert not np.abs(x.imag).any() and not np.abs(want.imag).any()
            x, want = x.real[0::2] + 1j * x.real[1::2], want.real[0::2] + 1j * want.real[1::2]
        dim = x.size
        internal = typ == 3 and swz >= 256

        def part(q):
            """(first reference index, rows, elements of the local array) of rank q"""
            if internal:
                a = [C.c_int64() for _ in range(4)]
                ck(L.dnm_vec_layout_partition(C.byref(sub), P, q, *[C.byref(t) for t in a]))
                return a[2].value, a[3].value, a[1].value
            qn, rem = divmod(dim, P)
            return q * qn + min(q, rem), qn + (1 if q < rem else 0), qn + (1 if q < rem else 0)

        def block_to_device(q):
            s0, rows, nloc = part(q)
            nat = upload(x[s0:s0 + rows])
            if internal:
                dst = dmalloc(16 * nloc)
                pq = B.Partition(q, P)
                ck(L.dnm_vec_layout_copy(C.byref(sub), C.byref(pq), dst, nat, 1, None))
                return dst
            if swz:
                dst = dmalloc(16 * nloc)
                ck(L.dnm_vec_swizzle_copy(dst, nat, nloc, swz, None))
                return dst
            return nat

        def block_to_host(q, p):
            s0, rows, nloc = part(q)
            if internal:
                nat = dmalloc(16 * rows)
                pq = B.Partition(q, P)
                ck(L.dnm_vec_layout_copy(C.byref(sub), C.byref(pq), nat, p, 0, None))
                return download(nat, rows)
            if swz:
                nat = dmalloc(16 * nloc)
                ck(L.dnm_vec_swizzle_copy(nat, p, nloc, swz, None))
                return download(nat, rows)
            return download(p, rows)

        mats = []
        for q in range(P):
            h = vp()
            pq = B.Partition(q, P)
            ck(L.dnm_mat_create(masks.size, B.p64(masks), B.p64(offs), B.p64(signs),
                                coeffs.view(np.float64).ctypes.data_as(B.f64p), C.byref(sub), C.byref(sub), 0, flags,
                                C.byref(pq), C.byref(h)))
            if exchange:
                chosen = C.c_int()
                ck(L.dnm_mat_set_exchange(h, exchange, C.byref(chosen)))
                assert chosen.value == exchange, (name, chosen.value)
            if masks.size and masks[0] == 0 and typ == 3:
                ck(L.dnm_mat_precompute_diagonal(h, None))        # the SpinConserve kernels read a cached diagonal
            mats.append(h)
        xs = [block_to_device(q) for q in range(P)]
        px = (vp * P)(*xs)
        pm = (vp * P)(*mats)
        errs = []
        for me in range(P):
            ck(L.dnm_comm_loopback(comm, me, P, px, pm))
            if me % 2 == 0:
                ck(L.dnm_comm_prepare(comm, mats[me], None))      # buffers / windows ahead of the first multiply (every other rank: lazily)
            s0, rows, nloc = part(me)
            y = upload(np.full(nloc, 5.0 + 1j, dtype=np.complex128))
            for _ in range(2):                                    # the second call runs on the cached plan
                ck(L.dnm_mat_mult_partitioned(mats[me], comm, xs[me], y, None))
            ck(L.dnm_stream_synchronize(None))
            got = block_to_host(me, y)
            errs.append(float(np.abs(got - want[s0:s0 + rows]).max()))
            ck(L.dnm_free(y))
        buf = C.create_string_buffer(4096)
        ck(L.dnm_mat_plan_describe(mats[0], buf, len(buf)))
        report["cases"][name] = {"P": P, "max_err": max(errs), "plan": buf.value.decode().strip().split("\n")[0][:100]}
        worst = max(worst, max(errs) / max(1.0, float(np.abs(want).max())))

        if name == os.environ.get("DNM_NATIVE_SOLVER_CASE", "sc3_P2"):
            # the solvers through the native hooks, on rank 0 of the partition (reductions are sums over one process:
            # what is checked is that the multiply of every step goes through dnm_mat_mult_partitioned and returns)
            ck(L.dnm_comm_loopback(comm, 0, P, px, pm))
            hooks = B.Hooks()
            ck(L.dnm_comm_hooks(comm, mats[0], None, C.byref(hooks)))
            s0, rows, nloc = part(0)
            yb = dmalloc(16 * nloc)
            ck(hooks.mult(hooks.ctx, xs[0], yb))
            ck(L.dnm_stream_synchronize(None))
            got = block_to_host(0, yb)
            assert float(np.abs(got - want[s0:s0 + rows]).max()) < 1e-11
            vals = (C.c_double * 2)(3.0, 4.0)
            ck(hooks.allreduce_sum(hooks.ctx, vals, 2))
            assert list(vals) == [3.0, 4.0]
            report["hooks"] = "ok"
        for kv in (str(g["env"]).split() if "env" in g else []):
            os.environ.pop(kv.split("=")[0])
        for q in range(P):
            ck(L.dnm_comm_forget(comm, mats[q]))
            ck(L.dnm_mat_destroy(mats[q]))
            ck(L.dnm_free(xs[q]))
    ck(L.dnm_comm_destroy(comm))
    report["worst_relative"] = worst
    print(json.dumps(report))
    return 0 if worst < 1e-12 else 1


if __name__ == "__main__":
    sys.exit(main())
