"""
A host in plain C against include/dynamite_amd.h (examples/c_abi_demo.c): the drop-in boundary is an extern "C" ABI with
plain pointers and sizes -- gcc compiles and links it with nothing but the header and the shared library.  On a GPU it
multiplies, checks Hermiticity and solves for the ground state of an XX chain against the filled Fermi sea; without one
the library's first call fails loudly (there is no CPU path).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    sys.path.insert(0, ROOT)
    from dynamite_amd import build as _b
    lib = _b.build()
    exe = os.path.join(str(tmp_path), "c_abi_demo")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"),
           "-o", exe, "-L" + os.path.dirname(lib), "-ldynamite_amd", "-lm", "-Wl,-rpath," + os.path.dirname(lib)]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_c_host_compiles_and_fails_loudly_without_a_gpu(tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: test_c_host_runs covers the run")
    out = subprocess.run([exe, "14"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "dnm_set_device" in out.stderr, (out.stdout, out.stderr)


@pytest.mark.gpu
def test_c_host_runs(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, "22"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout, out.stderr)
