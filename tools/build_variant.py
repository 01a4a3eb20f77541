"""
Builds an experimental variant of libdynamite_amd.so: ONE source recompiled with extra -D flags, linked with the
default build's other objects.  The variant libraries live under dynamite_amd/build/exp/ (git-ignored, shipped to
the GPU box by gpurun) and are picked at run time with DNM_EXPERIMENTAL=1 DNM_LIB=<path>.

  python tools/build_variant.py NAME [--src matvec_kernels.hip] -- -DDNM_XP_PRIO=1 ...
  python tools/build_variant.py --resources NAME     # VGPR / SGPR / scratch / occupancy of the variant's kernels
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamite_amd import build as B       # noqa: E402

EXP = os.path.join(B.HERE, "build", "exp")


def build_variant(name, src, extra, remarks=False, path=None):
    B.build()                                   # the default objects must exist
    os.makedirs(EXP, exist_ok=True)
    obj = os.path.join(EXP, "%s.%s.o" % (name, src))
    flags = ["--offload-arch=" + B.ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    cmd = [B._hipcc()] + flags + B.PER_FILE_FLAGS.get(src, []) + extra
    if remarks:
        cmd += ["-Rpass-analysis=kernel-resource-usage"]
    cmd += ["-I", B.CSRC, "-x", "hip", "-c", path or os.path.join(B.CSRC, src), "-o", obj]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = p.stdout.decode()
    if p.returncode:
        sys.stderr.write(out)
        raise SystemExit("hipcc failed")
    objs = [obj if s == src else os.path.join(B.HERE, "build", s + ".o") for s in B.SOURCES]
    lib = os.path.join(EXP, "lib_%s.so" % name)
    subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return lib, out


if __name__ == "__main__":
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    remarks = "--resources" in args
    args = [a for a in args if a != "--resources"]
    src = "matvec_kernels.hip"
    if "--src" in args:
        i = args.index("--src")
        src = args[i + 1]
        del args[i:i + 2]
    path = None
    if "--file" in args:                # compile this file in place of csrc/<src> (e.g. `git show HEAD:...` output)
        i = args.index("--file")
        path = args[i + 1]
        del args[i:i + 2]
    lib, out = build_variant(args[0], src, extra, remarks, path)
    print(lib)
    if remarks:
        import re
        cur = None
        want = os.environ.get("KERNELS", "tile_pass_kernelILi12ELi2ELb0ELi1|sc3_")
        for line in out.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
            if cur and re.search(want, cur) and re.search(r"VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size", line):
                print(cur[:60], line.split("remark:")[-1].strip())
