"""
The suite drives plan / kernel / solver knobs under DNM_EXPERIMENTAL=1 (tests/conftest.py) to reach at small sizes the
code that production reaches at large ones.  This file is the other half: what a USER's process computes -- no gate --
and that stray DNM_* variables in such a process change NOTHING: same plans, bit-identical vectors,
same eigenvalues as a process with a clean environment; each ignored knob is named once in a warning.  The clean
process's multiplies are checked against the oracle, so "the same" is also "right".
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.default_layout]

STRAY = {"DNM_TILE_BITS": "8", "DNM_LOG_ROWS": "3", "DNM_PLAN_MODE": "0", "DNM_GBITS": "2", "DNM_AMIN": "3", "DNM_SWZ": "6",
         "DNM_SC_LAYOUT": "6,4", "DNM_SC_SITE_PERM": "0", "DNM_SC3_TILED": "0", "DNM_SC3G_PTAB": "0", "DNM_EIGS_REAL": "0",
         "DNM_EIGS_BASISFREE": "1", "DNM_EXCHANGE": "partner", "DNM_CACHE_POLICY": "0", "DNM_WINDOW_FIRST": "0",
         "DNM_SC3G_KEEP_GATA": "0", "DNM_LIB": "/nonexistent/lib.so", "DNM_EXPM_ORTHO": "full", "DNM_DIAG_TABLE": "0",
         "DNM_TAB_RECORDS": "0", "DNM_DIAG_GROUPS": "0", "DNM_TAB_LOG_ROWS": "2", "DNM_SC_LAYOUT_MIN_DIM": "0", "DNM_SC_SOLVER_PARTITION": "0"}


def _child(extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("DNM_")}
    env.update(extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "production_env_child.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.fixture(scope="module")
def clean():
    return _child({})


def test_stray_knobs_change_nothing_in_a_production_process(clean):
    stray = _child(STRAY)
    assert clean["warnings"] == []
    assert clean["vec_swizzle"] == 16 and clean["sc_layout"] == [14, 10]           # the production layouts
    assert stray["vec_swizzle"] == 16 and stray["sc_layout"] == [14, 10]
    for name, c in clean["cases"].items():
        s = stray["cases"][name]
        assert s["plan"] == c["plan"], name
        assert (s["x"], s["y"], s["z"]) == (c["x"], c["y"], c["z"]), "%s: a stray knob changed a result bit" % name
        assert s["E0"] == c["E0"], name
    # the production plans are the ones DESIGN.md describes, not what the stray values ask for
    assert "B=12 logR=2 mode=2" in clean["cases"]["mbl_full_22"]["plan"]
    assert "internal layout [T 2 | W 10 | Lo 14]" in clean["cases"]["heisenberg_sc_26_13"]["plan"]
    assert "bond graph" in clean["cases"]["kagome_sc_27"]["plan"]
    assert "table records: " in clean["cases"]["syk_full_16"]["plan"]               # (DNM_TAB_RECORDS=0 among the stray ones)
    # every knob that was read is named (once) as ignored
    named = " ".join(stray["warnings"])
    for k in ("DNM_SWZ", "DNM_SC_LAYOUT", "DNM_SC_SITE_PERM"):
        assert k in named and "ignored" in named, stray["warnings"]


def test_the_production_process_is_right(clean):
    """... and the clean process's numbers are the oracle's: the same operators and seeded states rebuilt here."""
    from dynamite_amd import models
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import Full, SpinConserve
    from oracle import oracle as orc
    from gpu_util import orc_msc, orc_sub
    import hashlib
    for name, H, sub in (("mbl_full_22", models.mbl(22), Full(L=22)),
                         ("heisenberg_sc_26_13", models.heisenberg(26), SpinConserve(26, 13))):
        H.add_subspace(sub)
        x = State(L=H.L, subspace=sub, state='random', seed=7)
        xg = x.to_numpy()
        assert hashlib.sha256(np.ascontiguousarray(xg).tobytes()).hexdigest()[:24] == clean["cases"][name]["x"]
        y = H.dot(x).to_numpy()
        assert hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()[:24] == clean["cases"][name]["y"]
        ref = orc.matvec(orc_msc(H), orc_sub(sub), orc_sub(sub), xg, nthreads=8)
        assert np.max(np.abs(y - ref)) < 1e-12
        H.destroy_mat()
