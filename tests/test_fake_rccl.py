"""
The stand-in transport of the native-schedule tests (tests/fake_rccl/fake_rccl.cpp) checked on its own, on the CPU:
ranks are processes, buffers host memory (DNM_FAKE_RCCL_HOST=1).  It must match sends and receives per ordered pair of
ranks first in first out, let a pair send to each other inside one group, reduce and gather like the collectives it
stands for -- and REFUSE what real RCCL would silently corrupt or hang on: a receive of another size than the send, a
receive nobody sends to.  Exports the ten entry points csrc/comm.cpp binds (rccl_load).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
LIB = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")

ENTRY_POINTS = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart",
                "ncclGroupEnd", "ncclAllReduce", "ncclAllGather", "ncclGetErrorString"]


def _lib():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", LIB, SRC])
    return LIB


def test_exports_what_comm_cpp_binds():
    lib = C.CDLL(_lib())
    for name in ENTRY_POINTS:
        assert hasattr(lib, name), name
    bound = open(os.path.join(ROOT, "dynamite_amd", "csrc", "comm.cpp")).read()
    for name in ENTRY_POINTS:
        assert "DNM_SYM(%s)" % name[4:] in bound, name


CHILD = r'''
import ctypes as C, os, sys
import numpy as np
lib = C.CDLL(sys.argv[1])
rank, world, idfile, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
class Id(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Id, C.c_int]
lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ncclRecv.argtypes = lib.ncclSend.argtypes
lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
lib.ncclCommDestroy.argtypes = [C.c_void_p]
ident = Id()
ident.b = open(idfile, "rb").read()
comm = C.c_void_p()
assert lib.ncclCommInitRank(C.byref(comm), world, ident, rank) == 0
DOUBLE, INT64, SUM, MAX = 8, 4, 0, 2          # ncclDouble = ncclFloat64 = 8, ncclInt64 = 4; ncclSum = 0, ncclMax = 2
p = lambda a: a.ctypes.data_as(C.c_void_p)
rc = 0
if mode == "ok":
    # a ring inside ONE group, two messages per pair in each direction: first in, first out per ordered pair
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    a, b = np.full(5, 10.0 * rank + 1), np.full(3, 10.0 * rank + 2)
    ra, rb = np.zeros(5), np.zeros(3)
    assert lib.ncclGroupStart() == 0
    assert lib.ncclRecv(p(ra), 5, DOUBLE, prv, comm, None) == 0
    assert lib.ncclSend(p(a), 5, DOUBLE, nxt, comm, None) == 0
    assert lib.ncclSend(p(b), 3, DOUBLE, nxt, comm, None) == 0
    assert lib.ncclRecv(p(rb), 3, DOUBLE, prv, comm, None) == 0
    assert lib.ncclGroupEnd() == 0
    assert (ra == 10.0 * prv + 1).all() and (rb == 10.0 * prv + 2).all()
    v = np.array([rank + 1.0, -rank, 0.5])
    assert lib.ncclAllReduce(p(v), p(v), 3, DOUBLE, SUM, comm, None) == 0
    assert np.allclose(v, [world * (world + 1) / 2, -world * (world - 1) / 2, 0.5 * world])
    v = np.array([float(rank), -float(rank)])
    assert lib.ncclAllReduce(p(v), p(v), 2, DOUBLE, MAX, comm, None) == 0
    assert list(v) == [world - 1.0, 0.0]
    mine, allv = np.array([rank, rank * rank], dtype=np.int64), np.zeros(2 * world, dtype=np.int64)
    assert lib.ncclAllGather(p(mine), p(allv), 2, INT64, comm, None) == 0
    assert list(allv) == [x for q in range(world) for x in (q, q * q)]
elif mode == "size":
    # rank 0 sends 4 doubles, rank 1 expects 5: refused, not corrupted
    buf = np.zeros(5)
    if rank == 0:
        assert lib.ncclSend(p(buf), 4, DOUBLE, 1, comm, None) == 0
    elif rank == 1:
        rc = lib.ncclRecv(p(buf), 5, DOUBLE, 0, comm, None)
        assert rc != 0
        rc = 0
elif mode == "lost":
    # a receive nobody sends to: an error after the time limit, not a hang
    buf = np.zeros(2)
    if rank == 1:
        assert lib.ncclRecv(p(buf), 2, DOUBLE, 0, comm, None) != 0
assert lib.ncclCommDestroy(comm) == 0
sys.exit(rc)
'''


@pytest.mark.parametrize("mode,world", [("ok", 2), ("ok", 3), ("size", 2), ("lost", 2)])
def test_ranks_as_processes(tmp_path, mode, world):
    lib = C.CDLL(_lib())

    class Id(C.Structure):
        _fields_ = [("b", C.c_char * 128)]
    ident = Id()
    assert lib.ncclGetUniqueId(C.byref(ident)) == 0
    idfile = tmp_path / "id"
    idfile.write_bytes(bytes(ident.b).ljust(128, b"\0"))
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    env = dict(os.environ, DNM_FAKE_RCCL_HOST="1", DNM_FAKE_RCCL_TIMEOUT_S="3" if mode == "lost" else "60")
    procs = [subprocess.Popen([sys.executable, str(child), _lib(), str(r), str(world), str(idfile), mode], env=env,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    errs = [p.communicate(timeout=120)[1] for p in procs]
    assert [p.returncode for p in procs] == [0] * world, errs
    if mode == "size":
        assert "the sender posted 32 bytes, the receiver 40" in errs[1]
    if mode == "lost":
        assert "never arrived" in errs[1]
    name = bytes(ident.b).split(b"\0")[0].decode()
    assert not os.path.exists("/dev/shm/" + name), "mailboxes left behind"
