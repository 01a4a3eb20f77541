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


SAN_LIB = os.path.join(HERE, "libdynamite_amd_san.so")
# (UBSan reports and carries on, so that one run lists every finding; tools/sanitize.sh counts the reports)
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def sanitizer_runtime():
    """The AddressSanitizer runtime of ROCm's clang: to be LD_PRELOADed into the Python process that loads SAN_LIB."""
    import glob
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not hits:
        raise RuntimeError("no libclang_rt.asan-x86_64.so under /opt/rocm/lib/llvm")
    return hits[0]


def build_sanitized(force=False, verbose=False):
    """ASan + UBSan build of the HOST side of the library (CPU only, never for the GPU box): every source compiled
    with the sanitizers on its host side (`-Xarch_host`) -- planner, operator handles, the SpinConserve layout tables, the
    exchange schedules, the Krylov drivers and the kernels' host-side launch wrappers -- and linked into
    libdynamite_amd_san.so.  Host-only handles (DNM_MAT_HOST_ONLY) never launch, so the CPU tests run on it unchanged:
    `DNM_LIB_VARIANT=san LD_PRELOAD=<sanitizer_runtime()> python -m pytest ...` (tools/sanitize.sh)."""
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "dynamite_amd.h")]
    if not force and os.path.exists(SAN_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(SAN_LIB) for d in deps):
        return SAN_LIB
    objdir = os.path.join(HERE, "build", "san")
    os.makedirs(objdir, exist_ok=True)
    # sanitizers on the host side only (-Xarch_host): the kernel files still carry their device code, so that the
    # library registers its fat binaries as the product build does (GPU AddressSanitizer needs xnack+, which this pool
    # does not offer); the .cpp files have no device code at all
    flags = ["--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    for f in SAN_FLAGS:
        flags += [f] if f in ("-g", "-O1") else ["-Xarch_host", f]
    procs, objs = [], []
    for s in SOURCES:
        o = os.path.join(objdir, s + ".o")
        objs.append(o)
        cmd = [_hipcc()] + flags + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("== %s ==\n%s\n" % (s, out.decode()))
    if failed:
        raise RuntimeError("hipcc (sanitized host build) failed")
    subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-shared-libsan"] + SAN_FLAGS[:1] +
                          ["-o", SAN_LIB] + objs + ["-ldl"])
    return SAN_LIB


def build(force=False, verbose=False):
    if os.environ.get("DNM_LIB_VARIANT") == "san":
        return build_sanitized(force, verbose)
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
    if "--sanitize" in sys.argv:
        print(build_sanitized(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
