"""
Builds libdynamite_amd.so (hand-written HIP kernels + C ABI) in-tree with
hipcc for gfx950.  `python -m dynamite_amd.build` or `build()`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdynamite_amd.so")
SOURCES = ["matvec_kernels.hip", "sc3_kernels.hip", "sc3g_kernels.hip", "vec_kernels.hip", "rdm_kernels.hip", "plan.cpp", "sc3_perm.cpp", "mat.cpp", "vec_api.cpp", "krylov.cpp", "comm.cpp"]
ARCH = "gfx950"
# tile_pass_kernel sits at the 128-VGPR edge of 4 waves per SIMD; these two scheduler options of the AMDGPU backend
# measured -2.1 % on the L=30 multiply, same box (profiles/r02_exp22_sched.txt; max-ilp / iterative-minreg: +12...17 %);
# re-measured for the 4-rows-per-thread default (64 registers): 16.31-16.67 ms with them (the memory-clause strategy is
# the one that counts), 16.55-17.0 without or with max-ilp (profiles/r02_exp55_sched_rows4.txt)
PER_FILE_FLAGS = {"matvec_kernels.hip": ["-mllvm", "-amdgpu-schedule-relaxed-occupancy=true",
                                         "-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "dynamite_amd.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    # objects of sources that have left the library (they would ship to the GPU box with the snapshot)
    for f in os.listdir(objdir):
        if f.endswith(".o") and f[:-2] not in SOURCES:
            os.remove(os.path.join(objdir, f))
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    flags += os.environ.get("DNM_HIPCC_EXTRA", "").split()        # experiments (scheduler options ...)
    procs = []
    objs = []
    for s in SOURCES:
        o = os.path.join(objdir, s + ".o")
        objs.append(o)
        cmd = [_hipcc()] + flags + PER_FILE_FLAGS.get(s, []) + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("== %s ==\n%s\n" % (s, out.decode()))
        elif verbose and out:
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
