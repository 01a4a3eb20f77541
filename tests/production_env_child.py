"""
Child of tests/test_gpu_production_env.py: a PRODUCTION process -- no DNM_EXPERIMENTAL in its environment, whatever else
is -- builds operators the way a user does (dynamite_amd.models / Operator / State, no test helper, no knob), and prints
one JSON line: the plan of every operator, a digest of every result vector's bytes, eigenvalues, and the warnings the
knob gate raised.
"""
import hashlib
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    assert os.environ.get("DNM_EXPERIMENTAL") != "1"
    caught = []
    warnings.simplefilter("always")
    warnings.showwarning = lambda msg, *a, **k: caught.append(str(msg))
    import numpy as np
    from dynamite_amd import models
    from dynamite_amd.config import config
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import Full, Parity, SpinConserve, XParity

    def digest(state):
        return hashlib.sha256(np.ascontiguousarray(state.to_numpy()).tobytes()).hexdigest()[:24]
    out = {"vec_swizzle": config.vec_swizzle, "sc_layout": list(config.sc_layout or ()), "cases": {}}
    cases = [("mbl_full_22", models.mbl(22), Full(L=22)),
             ("xxz_parity_21", models.xxz(21), Parity('even', L=21)),
             ("heisenberg_sc_26_13", models.heisenberg(26), SpinConserve(26, 13)),        # internal layout: 10.4 M states
             ("kagome_sc_27", models.kagome("27b"), SpinConserve(27, 13)),                 # bond-graph passes, relabelled
             ("ising_xparity_20", models.ising(20), XParity(Full(L=20), sector='+')),
             ("syk_full_16", models.syk(16), Full(L=16)),                                  # table records
             ("long_range_full_18", models.long_range(18), Full(L=18))]                    # grouped diagonal terms
    for name, H, sub in cases:
        H.add_subspace(sub)
        x = State(L=H.L, subspace=sub, state='random', seed=7)
        y = H.dot(x)
        z = H.evolve(x, t=0.2)
        ev = H.eigsolve(nev=1, tol=1e-9, subspace=sub)
        out["cases"][name] = {"plan": H.get_mat(subspaces=(sub, sub)).describe().strip(), "y": digest(y), "z": digest(z),
                              "E0": float(ev[0]), "x": digest(x)}
        H.destroy_mat()
    out["warnings"] = sorted(set(w for w in caught if "DNM_" in w))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
