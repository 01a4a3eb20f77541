#!/usr/bin/env python
"""
Headline benchmark: matrix-free H|psi> for the random-field Heisenberg chain
(reference benchmarking/benchmark.py 'MBL' Hamiltonian, `--shell --mult`),
Full space, complex128, on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--L L]

A "step" is one multiply y = H x with x, y resident in HBM.  N=1: L=30 (2^30
amplitudes, 16 GiB per vector; BASELINE.json configs[2]).  N=2, 4: the state is
row-block partitioned, L = 30 + log2(N) so every GPU keeps 2^30 amplitudes
(weak scaling); N=8: L=34, the size of BASELINE.json configs[3] (2^31 amplitudes,
32 GiB per vector and GPU).  The rank exchange (partner blocks on 2 ranks, an
all-to-all between two layouts from 4 on -- DESIGN.md section 6) travels over
RCCL while the rank-local masks run.  Prints ONE JSON line on rank 0.

Every N > 1 line also carries, under `multi_gpu`, the multiply split three ways for BOTH schedules of the exchange -- the
native one (dnm_mat_mult_partitioned: the library's own RCCL communicator and stream; the default on RCCL) and the host
one (torch.distributed) --: the whole multiply, its messages alone (with the rate the busiest link reached,
`link_GBs_measured`, next to the assumed one the prediction uses), its kernels alone, and how much of the exchange the
schedule hid; the verdict of the first multiply's check against the MSC definition; and `secondary.config5`:
`eigsolve(nev=1)` of the Heisenberg chain in SpinConserve(36, 18) at N=8 (BASELINE.json configs[4]; (35,17) / (34,17) at
N = 4 / 2: the same rows per GPU).  Before a rank touches its GPU it sends a CHILD process through the native schedule
at a small size (`first_contact_probe`): a schedule that hangs or fails on hardware nobody has run it on costs that
child, and the measurement falls back to the host schedule and says so.  Every rank runs a watchdog thread: a
phase that makes no progress for `--watchdog` seconds prints the plan and the phase and ends the process with a
non-zero code (so a hung collective cannot hold the node); the parent of a self-launched run ends the other
ranks as soon as one fails.

Launch forms: under torch.distributed.run (RANK / WORLD_SIZE in the environment) every process is one
rank; a bare `python bench.py --gpus N` (N > 1, no WORLD_SIZE) starts its own N rank processes BEFORE
anything touches the GPU in the parent and forwards rank 0's line.  When fewer than N GPUs are visible the
ranks share the devices over gloo with host-staged blocks (a plumbing check, flagged "transport": "gloo");
with no GPU at all (`"dry_run": true`) only the plan, the exchange schedule and the gloo transport run.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_AMP = 32.0  # read x once + write y once
# what a two-launch plan must move per amplitude whatever the kernels do
FLOOR_BYTES_PER_AMP = 80.0
# what a device-to-device copy sustains on this chip (read + write, profiles/r03_copy_probe2.txt; 5.2-5.5 TB/s with the
# 64 KB-tile workgroup shape): the rate a memory-bound kernel can be held to, where 8 TB/s is the spec sheet
COPY_RATE_GBS = 6300.0
FLOOR_DERIVATION = ("two launches: x is read by both (2 x 16 B), y is written by the first, read back and written by "
                    "the second (3 x 16 B); one launch would have to keep all 29 bond exchanges of the 2^30 hypercube "
                    "on chip (DESIGN.md 4.2)")


def default_L(n_gpus):
    """N=1: BASELINE.json configs[2] (L=30).  N=2, 4: 2^30 amplitudes per GPU.  N=8: configs[3], L=34 (2^31 per GPU)."""
    return 34 if n_gpus == 8 else 30 + int(math.log2(n_gpus))


BASELINE_CONFIG = {(1, 30, "mbl"): "BASELINE.json configs[2]: L=30 random-field Heisenberg, 1 GPU",
                   (8, 34, "mbl"): "BASELINE.json configs[3] at its size (L=34, 2^34 amplitudes on 8 GPUs) with the "
                                   "headline's random fields kept: the same 33 bond masks and exchange as the plain "
                                   "Heisenberg chain, L more diagonal terms (--model heisenberg runs it without them)",
                   (8, 34, "heisenberg"): "BASELINE.json configs[3]: L=34 Heisenberg, 8 GPUs"}


def cpu_baseline(sample_L=26, reps=3):
    """Oracle (C restatement of the reference's MatMult_CPU_Fast: 2^11-row blocks, unswitched sum_term loops,
    contiguous-run do_cache_product, bpetsc_template_2.c:598-683, 713-889) timed on this box's host cores; a
    reported baseline, not the target.  Sample: L=26 (BASELINE.md section 5's CPU size) on all cores, plus a
    one-thread run at L=24 so that the per-core rate can be set against SURVEY section 6's anchor for the genuine
    reference C (4.3-6.8 Mamp/s per core); one multiply at the headline size L=30 is added when the host has the
    memory for it (DNM_BENCH_CPU_L30=0 / 1 forces the choice)."""
    import numpy as np
    from oracle import oracle as orc
    from dynamite_amd import models, msc_tools

    def time_one(L, nt, reps):
        H = models.mbl(L)
        H.reduce_msc()
        masks, offs = msc_tools.get_mask_offsets(H.msc)
        msc = orc.Msc(masks, offs, H.msc['signs'], H.msc['coeffs'])
        sub = orc.full(L)
        n = 1 << L
        x, out = np.empty(n, dtype=np.complex128), np.empty(n, dtype=np.complex128)
        orc.fill_test_vectors(x, out, nt)      # cheap non-trivial amplitudes; pages first touched by the workers
        best = float('inf')
        for _ in range(reps):
            t0 = time.perf_counter()
            orc.matvec(msc, sub, sub, x, nthreads=nt, out=out)
            best = min(best, time.perf_counter() - t0)
        return n / best, best

    ntmax = orc.max_threads()
    # the thread count that is fastest on this host (more threads than the container's CPU quota or the memory
    # system can feed are slower): probed at L=24, then used for the timed sample
    probe = {}
    for cand in sorted({ntmax, max(1, ntmax // 2), max(1, ntmax // 4), min(16, ntmax)}):
        probe[cand], _ = time_one(24, cand, 2)
    nt = max(probe, key=probe.get)
    rate, best = time_one(sample_L, nt, reps)
    rate1, best1 = time_one(24, 1, 1)
    n16 = min(16, ntmax)
    rate16 = probe.get(n16) or time_one(24, n16, 2)[0]
    out = {"value": rate / 1e9, "unit": "Gamplitudes/s", "cores": nt, "kind": "port", "host_threads_available": ntmax,
           "thread_probe_L24_Gamp_s": {str(k): v / 1e9 for k, v in sorted(probe.items())},
           "per_core_Mamp_s": rate / nt / 1e6, "one_thread_Mamp_s": rate1 / 1e6,
           "threads_%d_Mamp_s_per_thread" % n16: rate16 / n16 / 1e6,
           "reference_anchor_Mamp_s_per_core": [4.3, 6.8],
           "sample": f"L={sample_L} random-field Heisenberg, Full space, best of {reps} multiplies ({best:.2f} s each) "
                     f"of the oracle's MatMult_CPU_Fast restatement on {nt} OpenMP threads (the fastest of "
                     f"{sorted(probe)} probed at L=24; {ntmax} available) = {rate / nt / 1e6:.2f} "
                     f"Mamp/s per thread; one thread at L=24: {best1:.2f} s = "
                     f"{rate1 / 1e6:.2f} Mamp/s, {n16} threads at L=24: {rate16 / n16 / 1e6:.2f} Mamp/s per thread, against "
                     f"4.3-6.8 Mamp/s per core measured for the reference's own C (SURVEY section 6); the rate per "
                     f"thread falls with the thread count because every mask re-reads x from host memory "
                     f"(30 x 16 B per amplitude)"}
    def mem_available_gib():
        try:
            for ln in open("/proc/meminfo"):
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) / 2 ** 20
        except OSError:
            pass
        return 0.0

    l30 = os.environ.get("DNM_BENCH_CPU_L30")
    if l30 == "1" or (l30 is None and mem_available_gib() > 96 and rate > 0.05e9):     # 2 x 16 GiB, about 10-20 s
        try:
            r30, b30 = time_one(30, nt, 1)
            out["L30_Gamp_s"] = r30 / 1e9
            out["sample"] += f"; L=30, one multiply: {b30:.2f} s = {r30 / 1e9:.3f} Gamp/s"
        except MemoryError:
            pass
    return out


def secondary(wd, budget_s=36.0):
    """The Krylov half of the path (wall time and multiply count of evolve / eigsolve; the
    reference harness times them as phases of their own, benchmarking/benchmark.py:205-226, 311-313), on rank 0 of a
    one-GPU run, after the timed multiplies:
      * `evolve(t=1)` on the L=26 XXZ chain (BASELINE.json configs[1]);
      * the basis-free Lanczos of `eigsolve(nev=1)` at the headline size (L=30 random-field Heisenberg): ms per step
        and the step's own roofline -- one step is a multiply (32 B/amp) plus the three-term update sweep (48 B/amp:
        it reads v_j and w and writes v_{j+1}), `(32 + 48) * dim / t_step / peak`;
      * `eigsolve(nev=1)` on SpinConserve(32,16) (BASELINE.json configs[4]'s subspace family at one-GPU size), in
        complex128 and in the real arithmetic eigsolve takes on its own for a real-symmetric operator;
      * the reference's flagship example as its script runs it (examples/scripts/kagome/run_kagome.py:51-77): the
        30-site kagome torus in XParity(SpinConserve(30, 15)), `eigsolve(nev=2)` (round 5: bond-graph passes on a
        relabelled layout, real arithmetic).
      * the multiply of the harness's many-term operator, SYK at L=24 (table records, round 6), checked on sampled rows
        against the MSC definition.
    Each carries a sanity check (norm preserved / measured residual within tol).  A phase is skipped (and says so) once
    the budget is spent."""
    import numpy as np
    import torch
    from dynamite_amd import models, _lib
    from dynamite_amd.config import config
    from dynamite_amd.states import State
    from dynamite_amd.subspaces import Full, SpinConserve
    from dynamite_amd.computations import evolve, eigsolve
    t_begin = time.perf_counter()
    out = {}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    def left():
        return budget_s - (time.perf_counter() - t_begin)

    # -- evolve, L=26 XXZ
    wd.phase("secondary: evolve L=26")
    L = 26
    sub = Full(L=L)
    H = models.xxz(L)
    H.add_subspace(sub)
    psi = State(L=L, subspace=sub)
    psi.set_random(seed=0)
    res = State(L=L, subspace=sub)
    runs = []
    for _ in range(2):          # the first call grows the solver workspace
        _, dt = timed(lambda: H.evolve(psi, t=1.0, result=res))
        runs.append((dt, dict(evolve.last_stats)))
    nrm = res.norm()
    dt, st = runs[-1]
    out["evolve_L26_xxz_t1"] = {"wall_s": dt, "first_call_wall_s": runs[0][0], "matvecs": st["matvecs"],
                                "outer_steps": st["its"], "ms_per_matvec_equivalent": dt / max(1, st["matvecs"]) * 1e3,
                                "algo": "default: Krylov (expokit-style sub-steps) handing the rest of a real-time "
                                        "interval to the Chebyshev expansion",
                                "norm_error": abs(nrm - 1.0), "dim": 1 << L}
    if not abs(nrm - 1.0) < 1e-8:
        out["evolve_L26_xxz_t1"]["failed_checks"] = ["evolve did not preserve the norm: %r" % nrm]
    # -- the multiply of the same configuration (BASELINE.json configs[1]: L=26 XXZ, one GPU) on its own
    wd.phase("secondary: multiply L=26")
    mat = H.get_mat(subspaces=(sub, sub))
    xv, yv = psi.vec, res.vec
    for _ in range(3):
        mat.mult(xv, yv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nrep = 20
    torch.cuda.synchronize()
    e0.record()
    for _ in range(nrep):
        mat.mult(xv, yv)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / nrep
    gbs = ALG_BYTES_PER_AMP * (1 << L) / (ms * 1e-3) / 1e9
    r = {"ms": ms, "Gamplitudes_per_s": (1 << L) / (ms * 1e-3) / 1e9, "launches": mat.launches_per_mult(),
         "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
         "traffic_bytes_per_amp": None, "traffic_source": None, "plan_signature": plan_signature(mat), "dim": 1 << L,
         "baseline_config": "BASELINE.json configs[1]: L=26 XXZ chain, full space, 1 GPU"}
    try:        # counter bytes of the same plan on the same kernel sources, if the committed profile run holds them
        p = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json"))).get("config2") or {}
        if p.get("plan_signature") == r["plan_signature"] and p.get("kernel_source_hash") == kernel_source_hash():
            r["traffic_bytes_per_amp"] = p["bytes_per_amplitude"]
            r["traffic_source"] = "profiles/latest_pmc.json config2: rocprofv3 --pmc, separate run of `bench.py --L 26 --model xxz`"
    except Exception:       # noqa: BLE001
        pass
    out["multiply_L26_xxz"] = r
    del mat, xv, yv
    H.destroy_mat()
    del psi, res, H

    # -- one Lanczos step at the headline size, and evolve(t=1) there
    if left() > 7.0:
        wd.phase("secondary: Lanczos L=30")
        L = 30
        sub = Full(L=L)
        H = models.mbl(L)
        H.add_subspace(sub)
        tol = 1e-6
        dim = 1 << L
        # Every solve below is timed WARM: the basis-free Lanczos works in four vectors (64 GiB at L=30) that the
        # library keeps between solves; a first solve pays for acquiring them (fresh device memory is scrubbed when it
        # is handed out: 1-2 s).  They are acquired and touched here, untimed as a solve but timed on their own, so
        # that `wall_s` is what every solve after the first takes and `cold_wall_s` what the first one does.
        _, acquire_s = timed(lambda: _lib.check(_lib.lib().dnm_workspace_reserve(4 * dim * 16, None)))

        def lanczos(real):
            """complex128 vectors as the reference's EPS has them (real=False), then the path eigsolve takes on its
            own for this operator: real arithmetic, two amplitudes per complex128 element (half the bytes)"""
            config.eigs_real_arithmetic = real
            try:
                (ev, dt) = timed(lambda: H.eigsolve(nev=1, tol=tol))
                st = dict(eigsolve.last_stats)
            finally:
                config.eigs_real_arithmetic = None
            problems = []
            if not st["max_rel_residual"] <= tol * 1.01:
                problems.append("Lanczos residual %r above tol" % st["max_rel_residual"])
            if bool(st["real_arithmetic"]) != real:
                problems.append("arithmetic is not the one asked for")
            step_ms = dt / st["matvecs"] * 1e3
            per_amp = (ALG_BYTES_PER_AMP + 48.0) * (0.5 if real else 1.0)
            bw = per_amp * dim / (step_ms * 1e-3) / 1e9
            r = {"wall_s": dt, "call": "warm (workspace in place before the timed call)",
                 "workspace_acquire_s": acquire_s, "cold_wall_s": dt + acquire_s,
                 "matvecs": st["matvecs"], "E0": float(ev[0]),
                 "ms_per_step": step_ms, "measured_rel_residual": st["max_rel_residual"], "tol": tol,
                 "arithmetic": "real (f64, 8 B per amplitude)" if real else "complex128 (16 B per amplitude)",
                 "roofline": {"bound": "hbm", "alg_bytes_per_amp_per_step": per_amp, "achieved": bw,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bw / HBM_PEAK_GBS}}
            if problems:
                r["failed_checks"] = problems
            return r
        out["lanczos_L30"] = lanczos(False)
        if left() > 3.0:
            out["lanczos_L30_real_arithmetic"] = lanczos(True)
        if left() > 3.5:
            wd.phase("secondary: evolve L=30")
            psi = State(L=L, subspace=sub)
            psi.set_random(seed=0)
            res = State(L=L, subspace=sub)
            _, dt = timed(lambda: H.evolve(psi, t=1.0, result=res))
            st = dict(evolve.last_stats)
            nrm = res.norm()
            out["evolve_L30_mbl_t1"] = {"wall_s": dt, "call": "warm (workspace in place)", "matvecs": st["matvecs"],
                                        "outer_steps": st["its"],
                                        "ms_per_matvec_equivalent": dt / max(1, st["matvecs"]) * 1e3,
                                        "norm_error": abs(nrm - 1.0), "dim": dim}
            if not abs(nrm - 1.0) < 1e-8:
                out["evolve_L30_mbl_t1"]["failed_checks"] = ["evolve did not preserve the norm: %r" % nrm]
            del psi, res
        else:
            out["evolve_L30_mbl_t1"] = "skipped: budget"
        H.destroy_mat()
        del H
    else:
        out["lanczos_L30"] = "skipped: budget"

    # -- eigsolve on SpinConserve(32,16)
    if left() > 5.0:
        wd.phase("secondary: eigsolve SpinConserve(32,16)")
        L, k = 32, 16
        sub = SpinConserve(L, k)
        H = models.heisenberg(L)
        H.add_subspace(sub)
        tol = 1e-8

        def sc_solve(real):
            config.eigs_real_arithmetic = real
            try:
                (ev, dt) = timed(lambda: H.eigsolve(nev=1, tol=tol, subspace=sub))
            finally:
                config.eigs_real_arithmetic = None
            st = dict(eigsolve.last_stats)
            r = {"wall_s": dt, "call": "warm (its workspace is smaller than the one already in place)",
                 "matvecs": st["matvecs"], "E0": float(ev[0]), "dim": sub.get_dimension(),
                 "measured_rel_residual": st["max_rel_residual"], "tol": tol, "ms_per_step": dt / st["matvecs"] * 1e3,
                 "arithmetic": "real (f64, 8 B per amplitude)" if real else "complex128 (16 B per amplitude)"}
            problems = []
            if not st["max_rel_residual"] <= tol * 1.01:
                problems.append("residual %r above tol" % st["max_rel_residual"])
            if bool(st["real_arithmetic"]) != real:
                problems.append("arithmetic is not the one asked for")
            if problems:
                r["failed_checks"] = problems
            return r
        out["eigsolve_sc32_16"] = sc_solve(False)
        if left() > 2.0:
            out["eigsolve_sc32_16_real_arithmetic"] = sc_solve(True)      # what eigsolve takes on its own here
        H.destroy_mat()
    else:
        out["eigsolve_sc32_16"] = "skipped: budget"

    # -- the reference's flagship example: run_kagome.py 30 (ground state and gap in the Z2 sector)
    if left() > 6.0:
        wd.phase("secondary: kagome-30 eigsolve(nev=2)")
        from dynamite_amd.subspaces import XParity
        H = models.kagome("30")
        N = H.L
        sub = XParity(SpinConserve(N, N // 2), sector=-1)          # N % 4 == 2 (run_kagome.py:53-58)
        H.add_subspace(sub)
        (ev, dt) = timed(lambda: H.eigsolve(nev=2, subspace=sub))
        st = dict(eigsolve.last_stats)
        r = {"wall_s": dt, "call": "warm workspace; includes building the operator (site relabelling, tables)",
             "matvecs": st["matvecs"], "E0_per_site": float(ev[0]) / N, "gap": float(ev[1] - ev[0]),
             "dim": sub.get_dimension(), "measured_rel_residual": st["max_rel_residual"], "tol": 1e-8,
             "arithmetic": "real (f64, 8 B per amplitude)" if st["real_arithmetic"] else "complex128 (16 B per amplitude)",
             "plan": H.get_mat(subspaces=(sub, sub)).describe().strip().split(":")[0]}
        if not (abs(r["E0_per_site"] + 0.438477) < 1e-5 and st["max_rel_residual"] <= 1.01e-8):
            r["failed_checks"] = ["ground state energy per site %r (Lauchli et al.: -0.438477) or residual %r"
                                  % (r["E0_per_site"], st["max_rel_residual"])]
        out["eigsolve_kagome30_xparity_nev2"] = r
        H.destroy_mat()
    else:
        out["eigsolve_kagome30_xparity_nev2"] = "skipped: budget"
    # -- a known answer at config 5's subspace: 0.25 sum (XX + YY) on the open chain in SpinConserve(32, 16) is a chain of
    #    free fermions; its ground state the filled Fermi sea (tests/test_gpu_fullsize.py::test_xx_models_against_free_fermions)
    if left() > 3.0:
        wd.phase("secondary: XX chain against the Fermi sea")
        import numpy as _np
        from dynamite_amd.operators import sigmax, sigmay, op_sum
        Lx = 32
        H = op_sum(0.25 * (sigmax(i) * sigmax(i + 1) + sigmay(i) * sigmay(i + 1)) for i in range(Lx - 1))
        H.L = Lx
        sub = SpinConserve(Lx, Lx // 2)
        H.add_subspace(sub)
        (ev, dt) = timed(lambda: H.eigsolve(nev=1, tol=1e-9, subspace=sub))
        exact = float(_np.sort(_np.cos(_np.pi * _np.arange(1, Lx + 1) / (Lx + 1)))[:Lx // 2].sum())
        r = {"wall_s": dt, "E0": float(ev[0]), "exact": exact, "abs_error": abs(float(ev[0]) - exact),
             "dim": sub.get_dimension(), "matvecs": eigsolve.last_stats["matvecs"], "tol": 1e-9}
        if not r["abs_error"] < 1e-7:
            r["failed_checks"] = ["ground-state energy %r against the filled Fermi sea %r" % (r["E0"], exact)]
        out["known_answer_xx_chain_sc32_16"] = r
        H.destroy_mat()
    else:
        out["known_answer_xx_chain_sc32_16"] = "skipped: budget"
    # -- the many-term operator of the reference's harness: SYK (benchmarking/benchmark.py:139-160), L=24 -- 10 903 masks,
    #    194 580 terms; round 6: table records (csrc/plan.h DevTab).  Checked on sampled rows against the
    #    MSC definition evaluated in numpy (independent of every kernel).
    if left() > 12.0:
        wd.phase("secondary: SYK L=24 multiply")
        import numpy as _np
        Ls = 24
        t0 = time.perf_counter()
        H = models.syk(Ls)
        sub = Full(L=Ls)
        H.add_subspace(sub)
        mat = H.get_mat(subspaces=(sub, sub))
        build_s = time.perf_counter() - t0
        psi = State(L=Ls, subspace=sub)
        psi.set_random(seed=1)
        res = State(L=Ls, subspace=sub)
        xv, yv = psi.vec, res.vec
        for _ in range(2):
            mat.mult(xv, yv)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = 5
        torch.cuda.synchronize()
        e0.record()
        for _ in range(nrep):
            mat.mult(xv, yv)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / nrep
        msc = H.msc
        rows = _np.array([0, 1, (1 << Ls) - 1, 0x5a5a5a, 0x123456, 0xfedcba, 1 << 23, (1 << 12) + 7], dtype=_np.int64)
        mk, sg, cf = msc['masks'].astype(_np.int64), msc['signs'].astype(_np.int64), msc['coeffs']
        cols = rows[:, None] ^ mk[None, :]                   # the column state of every term, row by row
        par = _np.zeros(cols.shape, dtype=_np.int64)
        v = cols & sg[None, :]
        while _np.any(v):
            par ^= v & 1
            v >>= 1
        xs = xv.array[xv.positions(torch.from_numpy(cols.reshape(-1)).to(xv.array.device))].cpu().numpy().reshape(cols.shape)
        want = ((1 - 2 * par) * cf[None, :] * xs).sum(axis=1)
        got = yv.array[yv.positions(torch.from_numpy(rows).to(yv.array.device))].cpu().numpy()
        err, scale = float(_np.abs(got - want).max()), float(_np.abs(want).max())
        r = {"ms": ms, "dim": 1 << Ls, "nmasks": int(_np.unique(mk).size), "nterms": int(mk.size), "launches": mat.launches_per_mult(),
             "build_s": build_s, "includes": "build_s: the operator in Python (op_product of 194 580 Majorana strings) and its tables",
             "row_mask_pairs_per_s": float(_np.unique(mk).size) * (1 << Ls) / (ms * 1e-3),
             "sampled_rows_max_abs_error": err, "sampled_rows_scale": scale,
             "records": "table records (one look-up per row and mask; rounds 1-5 evaluated every term: 462 ms)",
             "plan_tail": mat.describe().strip().split("\n")[-1]}
        if not r["plan_tail"].startswith("table records:"):
            r["failed_checks"] = ["the operator did not take table records: %s" % r["plan_tail"]]
        if not err <= 1e-10 * max(scale, 1e-300):
            r.setdefault("failed_checks", []).append("sampled rows off by %r (scale %r)" % (err, scale))
        out["multiply_L24_syk"] = r
        del mat, xv, yv, psi, res
        H.destroy_mat()
    else:
        out["multiply_L24_syk"] = "skipped: budget"
    out["total_s"] = time.perf_counter() - t_begin
    return out


XGMI_LINK_GBS = 64.0      # assumed sustained one-direction rate of one xGMI link (spec 153 GB/s bidirectional per
                          # link pair; RCCL send/recv reaches 60-77 GB/s per direction): used for the PREDICTION only


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """Number of GPUs this process could use, WITHOUT initialising the HIP runtime (the parent of a self-launched
    run must stay clean of it: its children are the ones that touch the GPU).  KFD topology nodes with SIMDs are
    GPUs; *_VISIBLE_DEVICES narrows the set."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for ln in open(f):
                if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                    n += 1
        except OSError:
            pass
    if n == 0:
        # no readable KFD topology (or really no GPU): ask a short-lived child, which may initialise HIP and exit
        import subprocess
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                 capture_output=True, text=True, timeout=300)
            return int(out.stdout.strip().splitlines()[-1])
        except Exception:
            return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def spawn_ranks(n, timeout_s):
    """Parent of a bare `--gpus N` call: start N rank processes (this interpreter never touches the GPU) and
    supervise them: the first rank that fails, or the overall time limit, ends the others; exits with the worst
    return code."""
    import subprocess
    ndev = visible_gpus()
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if ndev < n:
        env["DNM_BENCH_BACKEND"] = "gloo"      # ranks share devices (or there is none): host-staged transport
        # (processes sharing a device: fewer hardware queues each, or four of them oversubscribe its queue slots and
        # every cross-stream wait costs a scheduler time slice -- tests/test_gpu_distributed.py::_worker)
        env.setdefault("GPU_MAX_HW_QUEUES", "2")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    t0 = time.monotonic()
    rc, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = abs(bad[0][1]) or 1, "rank %d exited with code %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if timeout_s and time.monotonic() - t0 > timeout_s:
            rc, why = 124, "time limit of %d s reached" % timeout_s
            break
        time.sleep(0.2)
    if why:
        sys.stderr.write("[bench] %s: ending the other ranks\n" % why)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 10:
            time.sleep(0.1)
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


class Watchdog:
    """A rank's own guard against a hang (a collective whose peer never arrives, a kernel that does not return):
    `phase(name)` marks progress; a phase older than `limit_s` prints what the rank was doing and the plan, then
    ends the process with code 3 -- termination only, never a re-exec of a process that holds the GPU."""

    def __init__(self, limit_s, rank):
        import threading
        self.limit, self.rank = limit_s, rank
        self.name, self.t, self.info = "start", time.monotonic(), ""
        self.done = []
        self.partial = None       # rank 0: the result line as far as it has been measured
        if limit_s > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def phase(self, name):
        self.done.append(self.name)
        self.name, self.t = name, time.monotonic()

    def _run(self):
        while True:
            time.sleep(min(1.0, self.limit / 4))
            if time.monotonic() - self.t > self.limit:
                sys.stderr.write("[bench watchdog] rank %d: no progress in phase '%s' for %.0f s; last completed: %s\n"
                                 "[bench watchdog] %s\n" % (self.rank, self.name, self.limit,
                                                            self.done[-1] if self.done else "-", self.info))
                sys.stderr.flush()
                if self.partial is not None:
                    # what has been measured is not lost to a later phase that hangs
                    try:
                        line = dict(self.partial)
                        line["aborted"] = {"phase": self.name, "reason": "no progress for %.0f s (watchdog)" % self.limit,
                                           "last_completed": self.done[-1] if self.done else None}
                        sys.stdout.write(json.dumps(line) + "\n")
                        sys.stdout.flush()
                    except Exception:       # noqa: BLE001
                        pass
                os._exit(3)


def exchange_estimate(summary, exchange_only_s=None):
    """Bytes this rank moves over xGMI per multiply (ShellMat.exchange_summary), the time the busiest link needs
    for them at the ASSUMED rate, and -- when the run timed the multiply's exchange on its own -- the rate the
    busiest link really reached (bytes on that link / time of the exchange alone): the number that confirms or
    refutes the prediction of DESIGN.md section 6."""
    out = {"exchange": summary["scheme"], "xgmi_bytes_per_step": int(summary["bytes_in"]),
           "xgmi_bytes_sent_per_step": int(summary["bytes_out"]), "xgmi_partners": int(summary["peers"]),
           "xgmi_busiest_link_bytes": int(summary["busiest_link_bytes"]), "xgmi_link_GBs_assumed": XGMI_LINK_GBS,
           "xgmi_link_bound_ms": summary["busiest_link_bytes"] / (XGMI_LINK_GBS * 1e9) * 1e3,
           "xgmi_exchange_only_ms": None, "xgmi_link_GBs_measured": None}
    if exchange_only_s and summary["busiest_link_bytes"]:
        out["xgmi_exchange_only_ms"] = exchange_only_s * 1e3
        out["xgmi_link_GBs_measured"] = summary["busiest_link_bytes"] / exchange_only_s / 1e9
    return out


def dry_run(args, world, rank, wd=None):
    """No GPU: build every rank's plan on the host, run the exchange schedule over gloo with host tensors and
    print the line with value = null.  Exercises launch, rendezvous, plan, schedule matching and transport."""
    import torch
    import torch.distributed as dist
    from dynamite_amd import models, backend, msc_tools, _lib
    from dynamite_amd.subspaces import Full
    L = args.L or (18 + int(math.log2(world)))
    if wd:
        wd.phase("dry run: plan L=%d" % L)
    H = models.BY_NAME[args.model](L)
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    sub = Full(L=L)
    lc, rc = sub._c(), sub._c()
    h = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], lc, rc, False, _lib.MAT_HOST_ONLY, rank, world)
    nloc = (1 << L) // world
    nbits = nloc.bit_length() - 1
    split = None
    if backend.use_transposed_exchange(world):
        split = backend.transpose_split(masks, offs, H.msc['signs'], H.msc['coeffs'], L, world, int(lc.vec_swizzle))
    if split is not None:
        # transposed exchange: both operators must plan without partner passes; the all-to-all runs both ways
        lo, hi, f = split
        plans = []
        for part in (lo, hi):
            hp = backend.create_mat(*part, lc, rc, False, _lib.MAT_HOST_ONLY, rank, world)
            assert backend.exchange_plan(hp) == ([], [])
            plans.append(C_describe(hp).strip().replace("\n", " | "))
            _lib.check(_lib.lib().dnm_mat_destroy(hp))
        pieces, own, cnt = backend.transpose_pieces(nbits, world.bit_length() - 1, f, rank)
        gidx = torch.arange(nloc, dtype=torch.float64) + float(rank * nloc)
        x = gidx.to(torch.complex128)
        xb = torch.empty_like(x)

        def one_step():
            for src, dst in ((x, xb), (xb, x)):          # there and back: x must come home unchanged
                if dst is x:
                    dst.fill_(-1.0)
                for r in backend.post_transpose(src, dst, pieces):
                    r.wait()
                for off in own:
                    dst[off:off + cnt] = src[off:off + cnt]
        summary = {"scheme": "transpose", "bytes_in": 2 * 16 * cnt * len(pieces), "bytes_out": 2 * 16 * cnt * len(pieces),
                   "peers": world - 1, "busiest_link_bytes": 2 * 16 * cnt * len(pieces) // (world - 1)}
    else:
        plans = None
        sends, recvs = backend.exchange_plan(h)
        x = torch.full((nloc,), complex(rank + 1, 0), dtype=torch.complex128)
        bufs = [torch.empty(cnt, dtype=x.dtype) for _, _, cnt in recvs]

        def one_step():
            for r in backend.post_exchange(x, sends, recvs, bufs):
                r.wait()
        per = {}
        for p_, _, c_ in recvs:
            per[p_] = per.get(p_, 0) + 16 * c_
        summary = {"scheme": "partner", "bytes_in": sum(per.values()), "bytes_out": sum(16 * c_ for _, _, c_ in sends),
                   "peers": len(per), "busiest_link_bytes": max(per.values()) if per else 0}
    if wd:
        wd.info = "plan: " + C_describe(h).strip().replace("\n", " | ")
        wd.phase("dry run: exchange steps")
    hang = float(os.environ.get("DNM_BENCH_TEST_HANG_RANK", "-1"))
    if hang == rank:            # tests: one rank never posts its exchange (what a lost peer looks like)
        time.sleep(10 ** 6)
    for _ in range(args.warmup):
        one_step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    dist.barrier()
    wall = time.perf_counter() - t0
    if split is not None:
        # layout B holds, at local index i' of rank q, the element whose global index has the two fields swapped
        g = torch.arange(nloc, dtype=torch.int64) + rank * nloc
        d = ((g >> f) ^ (g >> nbits)) & (world - 1)
        want_b = (g ^ (d << f) ^ (d << nbits)).to(torch.float64)
        ok = bool((xb.real == want_b).all()) and bool((x.real == gidx).all())
    else:
        ok = all(bool((b == complex(p + 1, 0)).all()) for (p, _, _), b in zip(recvs, bufs))
    t = torch.tensor([wall, 0.0 if ok else 1.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # config 5's exchange at a toy size: every rank's column window from its host-only handle, the ranges travel over gloo
    sec = None
    if not args.no_secondary:
        if wd:
            wd.phase("dry run: config 5's window exchange")
        try:
            sec = {"config5": config5_dry_run(world, rank, tuple(int(v) for v in args.config5.split(",")) if args.config5
                                              else (18, 9))}
        except Exception as e:       # noqa: BLE001
            sec = {"error": repr(e)}
    if rank == 0:
        buf = C_describe(h) if plans is None else "transposed exchange | A: " + plans[0] + " | B: " + plans[1]
        exch_s = float(t[0]) / args.steps
        est = exchange_estimate(summary, exch_s)
        host = schedule_entry("host", float("nan"), exch_s * 1e3, float("nan"), summary,
                              "exchange_ok: every message arrived where the schedule says" if t[1] == 0.0 else "failed")
        for k in ("ms_per_step", "compute_only_ms", "hidden_ms", "hidden_frac_of_the_shorter"):
            host[k] = None          # no kernel runs without a GPU
        sec_ok = sec is None or ("error" not in sec and sec["config5"].get("exchange_ok") is True)
        print(json.dumps({
            "metric": "matrix-free H|psi> Gamplitudes/s, random-field Heisenberg", "value": None,
            "unit": "Gamplitudes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": float(t[0]) * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic", "dry_run": True,
            "transport": "gloo", "exchange_ok": bool(t[1] == 0.0),
            "config": dict({"workload": f"DRY RUN (no GPU): plan + exchange schedule of the L={L} random-field "
                                        f"Heisenberg chain on {world} ranks, no multiply executed",
                            "L": L, "schedule": "host (torch.distributed)", "exchange_selfcheck": None,
                            "plan": buf.strip().replace("\n", " | ")}, **est),
            "multi_gpu": {"default_schedule": "host", "first_contact_probe": None, "xgmi_link_GBs_assumed": XGMI_LINK_GBS,
                          "schedules": {"host": host, "native": "not run: no GPU (dnm_mat_mult_partitioned launches kernels)"}},
            "secondary": sec, "secondary_ok": sec_ok}))
    _lib.check(_lib.lib().dnm_mat_destroy(h))
    dist.destroy_process_group()
    sys.exit(0 if (t[1] == 0.0 and (sec is None or "error" not in sec)) else 1)


def config5_dry_run(world, rank, Lk):
    """Without a GPU: the exchange of config 5's multiply -- the Heisenberg chain in SpinConserve(L, k), internal
    three-field layout, whole blocks of equal top bits per rank -- planned by every rank from a host-only handle (its
    column window and the ranges of it its rows read), posted over gloo as ShellMat._mult_window posts it, and checked:
    every position a rank reads must hold what its owner holds there."""
    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    from dynamite_amd import models, backend, msc_tools, _lib
    from dynamite_amd.subspaces import SpinConserve
    L, k = Lk
    sub = SpinConserve(L, k)
    a, w = (14, 10) if L >= 28 else (6, 4)
    H = models.heisenberg(L)
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)

    def plan(order):
        """(descriptor, handle, window, exact needed ranges) of this rank in one block order of the layout"""
        dd = _lib.Subspace.from_buffer_copy(sub._c())
        dd.vec_swizzle = a | (w << 8) | (order << 16)
        hh = backend.create_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], dd, dd, False, _lib.MAT_HOST_ONLY, rank, world)
        lo, hi = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().dnm_mat_column_window(hh, C.byref(lo), C.byref(hi), None))
        nr = C.c_int64()
        _lib.check(_lib.lib().dnm_mat_column_ranges(hh, 0, None, C.byref(nr)))
        rg = (C.c_int64 * (2 * max(1, nr.value)))()
        _lib.check(_lib.lib().dnm_mat_column_ranges(hh, nr.value, rg, C.byref(nr)))
        return dd, hh, (lo.value, hi.value), [(int(rg[2 * i]), int(rg[2 * i + 1])) for i in range(nr.value)]
    # the reference-compatible partition (planned only: its volume beside the other's) ...
    d0, h0, _, needs0 = plan(0)
    i00, il0 = backend.layout_partition(d0, world, rank)[:2]
    remote0 = sum(max(0, min(b_, i00) - a_) + max(0, b_ - max(a_, i00 + il0)) for a_, b_ in needs0)
    _lib.check(_lib.lib().dnm_mat_destroy(h0))
    # ... and the one made for the exchange, which eigsolve(getvecs=False) takes on several ranks: posted for real
    d, h, window, needs = plan(1)
    allw = [None] * world
    dist.all_gather_object(allw, (window, needs))
    windows, allneeds = [v[0] for v in allw], [v[1] for v in allw]
    owned = [backend.layout_partition(d, world, q)[:2] for q in range(world)]
    i0, il = owned[rank]
    x = (torch.arange(il, dtype=torch.float64) + float(i0)).to(torch.complex128)     # every position holds its own number
    buf = backend.exchange_window(x, owned, windows, rank, None, allneeds)
    ok = True
    for a_, b_ in needs:
        ok = ok and bool((buf[a_ - window[0]:b_ - window[0]].real == torch.arange(a_, b_, dtype=torch.float64)).all())
    recvs, sends = backend.window_exchange_ops(owned, windows, rank, allneeds)
    t = torch.tensor([0.0 if ok else 1.0, float(sum(16 * (b_ - a_) for _, a_, b_ in recvs)), 16.0 * remote0],
                     dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    _lib.check(_lib.lib().dnm_mat_destroy(h))
    return {"workload": "DRY RUN: window exchange of the Heisenberg chain in SpinConserve(%d,%d) on %d ranks, layout (%d,%d), "
                        "on the partition made for the exchange (block order 1)" % (L, k, world, a, w),
            "dim": sub.get_dimension(), "exchange": "window", "exchange_ok": bool(t[0] == 0.0),
            "bytes_received_per_multiply_busiest_rank": int(t[1]),
            "bytes_received_per_multiply_busiest_rank_reference_compatible_partition": int(t[2]),
            "window_bytes_rank0": 16 * (window[1] - window[0] + 1),
            "ranges_read_rank0": len(needs), "heisenberg": None, "known_answer_xx_chain": None}


def plan_signature(mat):
    """Identifies the executed plan in profiles/latest_pmc.json (the counter run must be of the same plan)."""
    import hashlib
    return hashlib.sha256(mat.describe().strip().replace("\n", " | ").encode()).hexdigest()[:16]


def kernel_source_hash():
    """Hash of the sources the headline kernel and its plan come from: profiles/latest_pmc.json carries the hash of
    the tree its counters were taken on, and the line's `traffic` is null when the code has moved since."""
    import hashlib
    h = hashlib.sha256()
    for f in ("dynamite_amd/csrc/matvec_kernels.hip", "dynamite_amd/csrc/plan.cpp", "dynamite_amd/csrc/plan.h",
              "dynamite_amd/csrc/kernels.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def C_describe(handle):
    import ctypes as C
    from dynamite_amd import _lib
    buf = C.create_string_buffer(8192)
    _lib.check(_lib.lib().dnm_mat_plan_describe(handle, buf, len(buf)))
    return buf.value.decode()


# ---------------------------------------------------------------------------------------------------------------
# several GPUs
# ---------------------------------------------------------------------------------------------------------------
CONFIG5_BY_WORLD = {2: (34, 17), 4: (35, 17), 8: (36, 18)}      # about 1.13-1.17 G rows per GPU each; 8: BASELINE configs[4]
PROBE_PORT_OFFSET = 23


def first_contact_probe(args, world, rank, which="native"):
    """Send a CHILD process (one per rank, its own rendezvous on MASTER_PORT + offset, started before this process
    touches the GPU) through one schedule of the exchange at a small size: a partitioned Full-space multiply checked
    against the MSC definition, its exchange alone, and an eigsolve on a partitioned SpinConserve subspace through the
    solver hooks.  A schedule that hangs or fails on a machine nobody has run it on then costs that child -- ended by
    exact PID after `--probe-timeout` seconds -- and not the measurement.  Returns {"ok": bool, ...}."""
    import subprocess
    import tempfile
    # (under torch.distributed.run the ranks would look for the agent's store: the children make their own rendezvous)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + PROBE_PORT_OFFSET + (0 if which == "native" else 1))
    env["DNM_NATIVE_COMM"] = "1" if which == "native" else "0"
    out = tempfile.NamedTemporaryFile(prefix="dnm_probe_%d_" % rank, suffix=".json", delete=False)
    out.close()
    cmd = [sys.executable, os.path.abspath(__file__), "--probe-child", out.name, "--gpus", str(world),
           "--watchdog", str(max(30, args.probe_timeout - 20))]
    t0 = time.monotonic()
    res = {"ok": False, "schedule": which, "timeout_s": args.probe_timeout}
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL)     # (the report comes through the file)
        try:
            rc = p.wait(timeout=args.probe_timeout)
            res["returncode"] = rc
        except subprocess.TimeoutExpired:
            p.kill()                        # this exact child
            p.wait()
            res["error"] = "no answer within %d s: ended" % args.probe_timeout
            rc = None
        if rc == 0:
            try:
                res.update(json.load(open(out.name)))
            except Exception as e:       # noqa: BLE001
                res["error"] = "no report: %r" % (e,)
        elif rc is not None:
            res["error"] = "exit code %d" % rc + (" (the child's own watchdog: a phase made no progress)" if rc == 3 else "")
    except Exception as e:       # noqa: BLE001
        res["error"] = repr(e)
    finally:
        try:
            os.unlink(out.name)
        except OSError:
            pass
    res["wall_s"] = time.monotonic() - t0
    return res


def probe_child(args):
    """The child of `first_contact_probe`: a rank of a small run through the schedule DNM_NATIVE_COMM names."""
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0"))
    wd = Watchdog(args.watchdog, rank)
    wd.phase("probe: import torch")
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    backend_name = os.environ.get("DNM_BENCH_BACKEND", "nccl")
    wd.phase("probe: init_process_group(%s)" % backend_name)
    if backend_name == "nccl":
        from dynamite_amd import _comm
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=_comm.nccl_options())
    else:
        dist.init_process_group(backend_name)
    from dynamite_amd import models, backend
    from dynamite_amd.config import config
    from dynamite_amd.subspaces import Full, SpinConserve
    from dynamite_amd.states import State
    from dynamite_amd.computations import eigsolve
    config._initialize()
    rep = {"ok": False, "native": backend.native_transport()}
    L = args.L or (22 + int(math.log2(world)))
    wd.phase("probe: multiply L=%d" % L)
    H = models.mbl(L)
    sub = Full(L=L)
    H.add_subspace(sub)
    mat = H.get_mat(subspaces=(sub, sub))
    x, y = mat.createVecs()
    x.set_random(0)
    for _ in range(2):
        mat.mult(x, y)          # (the first one of a transposed exchange checks itself)
    err, scale = mat.selfcheck(x, y)
    rep["multiply_selfcheck"] = {"max_abs_dev": err, "scale": scale, "scheme": mat.exchange_summary()["scheme"]}
    assert err <= 1e-9 * max(scale, 1e-300), "sampled rows of the multiply are off by %r" % err
    wd.phase("probe: exchange alone")
    mat.exchange_only(x, y)
    wd.phase("probe: eigsolve on a partitioned SpinConserve subspace")
    Ls = 26
    ssub = SpinConserve(Ls, Ls // 2)
    Hs = models.heisenberg(Ls)
    Hs.add_subspace(ssub)
    ev = Hs.eigsolve(nev=1, tol=1e-8, subspace=ssub)
    st = dict(eigsolve.last_stats)
    rep["eigsolve_sc26_13"] = {"E0": float(ev[0]), "matvecs": st["matvecs"], "rel_residual": st["max_rel_residual"]}
    # (the Bethe-ansatz energy per site of the infinite chain is 1/4 - ln 2 = -0.4431: an open chain of 26 sits a little above)
    assert st["max_rel_residual"] <= 1.01e-8 and -0.4432 * Ls < float(ev[0]) < -0.40 * Ls, rep
    torch.cuda.synchronize()
    dist.barrier()
    rep["ok"] = True
    json.dump(rep, open(args.probe_child, "w"))
    Hs.destroy_mat()
    H.destroy_mat()
    backend.release_native_comm()
    dist.destroy_process_group()
    return 0


def phase_times(mat, x, y, steps, barrier, reduce_max):
    """One schedule of a partitioned multiply split three ways: (whole multiply, messages alone, kernels alone) in ms,
    each the max over ranks of the mean over `steps` calls between barriers."""
    def run(fn, n):
        fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        return reduce_max((time.perf_counter() - t0) / n) * 1e3
    whole = run(lambda: mat.mult(x, y), steps)
    exch = run(lambda: mat.exchange_only(x, y), max(1, min(3, steps)))
    comp = run(lambda: mat.compute_only(x, y), max(1, min(5, steps)))
    return whole, exch, comp


def schedule_entry(name, whole_ms, exch_ms, comp_ms, summary, selfcheck):
    hidden = exch_ms + comp_ms - whole_ms
    busiest = summary["busiest_link_bytes"]
    return {"schedule": name, "ms_per_step": whole_ms, "exchange_only_ms": exch_ms, "compute_only_ms": comp_ms,
            "hidden_ms": hidden, "hidden_frac_of_the_shorter": hidden / max(1e-9, min(exch_ms, comp_ms)),
            "busiest_link_bytes": int(busiest),
            "link_GBs_measured": (busiest / (exch_ms * 1e-3) / 1e9) if busiest else None,
            "selfcheck": selfcheck}


def config5(wd, world, rank, Lk=None, tol=1e-8):
    """BASELINE.json configs[4]: `eigsolve(nev=1)` of the Heisenberg chain in SpinConserve(36, 18) on 8 GPUs
    (computations.py:128-292 on bsubspace_impl.h:161-261 in the reference); (35,17) / (34,17) on 4 / 2 GPUs keep the rows
    per GPU.  Column windows over the ranks, the real arithmetic eigsolve takes on its own for a real-symmetric operator,
    reductions and multiplies through the solver hooks.  Then the same subspace's known answer: 0.25 sum (XX + YY) against the
    filled Fermi sea."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from dynamite_amd import models
    from dynamite_amd.subspaces import SpinConserve
    from dynamite_amd.computations import eigsolve
    from dynamite_amd.operators import sigmax, sigmay, op_sum
    L, k = Lk or CONFIG5_BY_WORLD[world]
    out = {"workload": "eigsolve(nev=1, tol=%g), Heisenberg chain, SpinConserve(%d,%d) on %d GPUs" % (tol, L, k, world),
           "baseline_config": "BASELINE.json configs[4]" if (L, k, world) == (36, 18, 8) else
                              "configs[4]'s family at this rank count (rows per GPU kept)"}
    sub = SpinConserve(L, k)
    out["dim"] = sub.get_dimension()

    def solve(H, name):
        wd.phase("config 5: %s" % name)
        H.add_subspace(sub)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev = H.eigsolve(nev=1, tol=tol, subspace=sub)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = dict(eigsolve.last_stats)
        # the handle the solve multiplied with: eigenvalues alone run on the partition made for the exchange
        # (Operator.get_solver_mat: T blocks in an order whose contiguous ranges cut one bond of the chain)
        mat = eigsolve.last_mat
        summ = mat.exchange_summary()
        ref = (H.get_real_packed_mat(sub) if st["real_arithmetic"] else H.get_mat(subspaces=(sub, sub))).exchange_summary()
        # rank 0 is an end of the partition and receives least: what a link-bound multiply waits for is the busiest rank / link
        mx = torch.tensor([float(summ["bytes_in"]), float(summ["busiest_link_bytes"]), float(ref["bytes_in"]),
                           float(ref["busiest_link_bytes"])], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        r = {"wall_s": dt, "includes": "building the operator (tables, windows of all ranks) and the solve",
             "matvecs": st["matvecs"], "ms_per_step": dt / max(1, st["matvecs"]) * 1e3, "E0": float(ev[0]),
             "measured_rel_residual": st["max_rel_residual"], "tol": tol,
             "arithmetic": "real (f64, 8 B per amplitude)" if st["real_arithmetic"] else "complex128 (16 B per amplitude)",
             "rows_this_rank": int(mat.m_local) * (2 if st["real_arithmetic"] else 1),
             "partition": ("made for the exchange (block order %d of the layout: csrc/sc3.h)" % (mat.swz_right >> 16)
                           if mat.swz_right >= (1 << 16) else "reference-compatible (ranges of the reference order)"),
             "bytes_received_per_multiply_rank0_reference_compatible_partition": int(ref["bytes_in"]),
             "exchange": summ["scheme"], "bytes_received_per_multiply_rank0": int(summ["bytes_in"]),
             "busiest_link_bytes_rank0": int(summ["busiest_link_bytes"]),
             "bytes_received_per_multiply_busiest_rank": int(mx[0].item()), "busiest_link_bytes_any_rank": int(mx[1].item()),
             "bytes_received_per_multiply_busiest_rank_reference_compatible_partition": int(mx[2].item()),
             "busiest_link_bytes_any_rank_reference_compatible_partition": int(mx[3].item()),
             "window_bytes_rank0": int(summ.get("window_bytes", 0)),
             "plan": mat.describe().strip().split("\n")[0][:160]}
        if not st["max_rel_residual"] <= tol * 1.01:
            r["failed_checks"] = ["residual %r above tol" % st["max_rel_residual"]]
        # the multiply of the solve on its own, split as the Full-space one above: whole / messages alone / kernels alone
        try:
            wd.phase("config 5: %s, phases of the multiply" % name)
            from dynamite_amd.backend import RawVec
            xr = RawVec(torch.randn(2 * mat.n_local, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
                        .view(torch.complex128), mat.swz_right)
            yr = RawVec(torch.zeros(mat.m_local, dtype=torch.complex128, device=xr.array.device), mat.swz_left)
            xr.perm, yr.perm = mat.perm_right, mat.perm_left

            def barrier():
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()

            def reduce_max(v):
                t = torch.tensor([v], dtype=torch.float64, device=xr.array.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            w_, e_, c_ = phase_times(mat, xr, yr, 3, barrier, reduce_max)
            r["multiply"] = {"ms": w_, "exchange_only_ms": e_, "compute_only_ms": c_, "hidden_ms": e_ + c_ - w_,
                             "schedule": "native" if mat._native_in_use(xr) else "host",
                             "link_GBs_measured_rank0": (summ["busiest_link_bytes"] / (e_ * 1e-3) / 1e9) if summ["busiest_link_bytes"] else None}
            del xr, yr
        except Exception as e:       # noqa: BLE001
            r["multiply"] = {"error": repr(e)}
        H.destroy_mat()
        return r
    out["heisenberg"] = solve(models.heisenberg(L), "Heisenberg chain")
    Hx = op_sum(0.25 * (sigmax(i) * sigmax(i + 1) + sigmay(i) * sigmay(i + 1)) for i in range(L - 1))
    Hx.L = L
    r = solve(Hx, "XX chain against the filled Fermi sea")
    exact = float(np.sort(np.cos(np.pi * np.arange(1, L + 1) / (L + 1)))[:k].sum())
    r["exact"], r["abs_error"] = exact, abs(r["E0"] - exact)
    if not r["abs_error"] < 1e-6 * abs(exact):
        r.setdefault("failed_checks", []).append("ground-state energy %r against the filled Fermi sea %r" % (r["E0"], exact))
    out["known_answer_xx_chain"] = r
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--model", default="mbl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the evolve / eigsolve phases of the line")
    ap.add_argument("--watchdog", type=int, default=900, help="seconds a phase may take before the rank gives up (0: off)")
    ap.add_argument("--timeout", type=int, default=3600, help="time limit of a self-launched multi-rank run (0: none)")
    ap.add_argument("--no-probe", action="store_true", help="several GPUs: skip the first-contact probe of the native schedule")
    ap.add_argument("--probe-timeout", type=int, default=300)
    ap.add_argument("--probe-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--config5", default=None, help="L,k of the SpinConserve eigsolve of a multi-GPU line (default by rank count)")
    args = ap.parse_args()

    if args.probe_child:
        sys.exit(probe_child(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, args.timeout)          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    wd = Watchdog(args.watchdog if world > 1 else 0, rank)
    wd.phase("import torch")
    import torch
    if world > 1 and torch.cuda.device_count() == 0:
        import torch.distributed as dist
        wd.phase("init_process_group(gloo)")
        dist.init_process_group("gloo")
        dry_run(args, world, rank, wd)      # does not return
    probe = None
    if world > 1:
        import torch.distributed as dist
        backend_name = os.environ.get("DNM_BENCH_BACKEND", "nccl")
        # The native schedule has never met a second device: before this process touches its GPU, a child goes through it
        # at a small size (torch.cuda.device_count() above does not initialise the runtime).  DNM_NATIVE_COMM=0 / 1 is a
        # decision already taken: no probe.
        # (DNM_BENCH_FORCE_PROBE=1: tests send the probe through a stand-in transport -- DNM_RCCL_LIB -- on gloo-staged ranks)
        if ((backend_name == "nccl" or os.environ.get("DNM_BENCH_FORCE_PROBE") == "1")
                and os.environ.get("DNM_NATIVE_COMM", "") == "" and not args.no_probe):
            wd.phase("first-contact probe of the native schedule (child process)")
            wd.t += args.probe_timeout          # (the probe has its own limit)
            probe = first_contact_probe(args, world, rank, "native")
        # rank % device count, as the reference picks its GPU (bcuda_template_2.cu:64-67)
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        # DNM_BENCH_BACKEND=gloo: dry run of the multi-rank flow with several ranks on one GPU (blocks staged
        # through the host; RCCL refuses two ranks on one device) -- a plumbing check, not a measurement
        wd.phase("init_process_group(%s)" % backend_name)
        if backend_name == "nccl":
            from dynamite_amd import _comm
            # (RCCL's stream with high priority: without it the exchange and the kernels it should hide under serialise on
            # this system -- _comm.nccl_options)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=_comm.nccl_options())
        else:
            dist.init_process_group(backend_name)
    from dynamite_amd import models, backend, _lib
    from dynamite_amd.config import config
    from dynamite_amd.subspaces import Full
    config._initialize()
    if probe is not None:
        # every rank or none
        t = torch.tensor([1.0 if probe["ok"] else 0.0], dtype=torch.float64, device=config.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        probe["all_ranks_ok"] = bool(t.item() == 1.0)
        if probe["all_ranks_ok"] and dist.get_backend() != "nccl":
            config.native_comm = True           # (a forced probe on gloo-staged ranks: the stand-in transport it went through)
        if not probe["all_ranks_ok"]:
            config.native_comm = False
            if rank == 0:
                print("[bench] first-contact probe of the native schedule failed (%s): host schedule over torch.distributed"
                      % probe.get("error", "on another rank"), file=sys.stderr)

    n_gpus = world
    L = args.L or default_L(n_gpus)
    wd.phase("build_mat L=%d" % L)
    H = models.BY_NAME[args.model](L)
    H.establish_L()
    H.reduce_msc()
    from dynamite_amd import msc_tools
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    sub = Full(L=L)

    def build(exchange=None):
        return backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c(), exchange=exchange)
    mat = build()
    if rank == 0:
        print(mat.describe(), file=sys.stderr)
    wd.info = "plan: " + mat.describe().strip().replace("\n", " | ")
    dim = 1 << L
    wd.phase("allocate and fill the vectors")
    x, y = mat.createVecs()
    x.set_random(0)
    x.normalize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=config.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def warm(mat, n, what):
        """n warm-up multiplies; the first multiply of a transposed-exchange operator checks sampled rows (collective):
        on a disagreement every rank sees the same verdict and the operator is rebuilt on partner blocks.  Returns
        (matrix, verdict)."""
        verdict = None
        for i in range(n):
            wd.phase("%s: warm-up multiply %d of %d" % (what, i + 1, n))
            try:
                mat.mult(x, y)
            except backend.ExchangeCheckError as e:
                verdict = "failed: " + str(e)
                if rank == 0:
                    print("[bench] " + str(e), file=sys.stderr)
                mat.destroy()
                mat = build(exchange='partner')
                mat.mult(x, y)
            if world > 1:
                torch.cuda.synchronize()        # so that the watchdog names the multiply that hangs
        if world > 1 and verdict is None:
            if mat.exchange_summary()["scheme"] == "transpose":
                verdict = "sampled rows of the first multiply agree with the MSC definition"
            else:
                # (partner blocks need no check of their own; on a first contact they get one all the same)
                err, scale = mat.selfcheck(x, y)
                verdict = ("sampled rows agree with the MSC definition" if err <= 1e-9 * max(scale, 1e-300)
                           else "failed: sampled rows off by %.3e (scale %.3e)" % (err, scale))
        return mat, verdict

    mat, selfcheck = warm(mat, max(1, args.warmup) if world > 1 else args.warmup, "default schedule")
    launches = mat.launches_per_mult()
    native_default = world > 1 and mat._native_in_use(x)
    wd.phase("barrier before the timed steps")
    barrier()
    wd.phase("%d timed multiplies" % args.steps)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        mat.mult(x, y)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=config.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])
    ms_per_step = wall * 1e3 / args.steps
    # sanity: the timed multiply produced a finite vector of the expected size
    ynorm = y.norm()
    assert math.isfinite(ynorm) and ynorm > 0
    summary = mat.exchange_summary()
    if world > 1:       # the heaviest rank (partner blocks differ from rank to rank: rank 0 needs the least)
        t = torch.tensor([summary["bytes_in"], summary["bytes_out"], summary["busiest_link_bytes"], summary["peers"]],
                         dtype=torch.float64, device=config.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        summary.update(bytes_in=int(t[0]), bytes_out=int(t[1]), busiest_link_bytes=int(t[2]), peers=int(t[3]))

    out = None
    if rank == 0:
        # kernel time from HIP events on the launch stream (events bracket K steps
        # of back-to-back launches; per-launch = total / (K * launches per step))
        kern_ms = dev_ms / args.steps
        avg_launch_ms = kern_ms / launches
        dim_local = dim // n_gpus
        alg_bytes_launch = ALG_BYTES_PER_AMP * dim_local / launches
        achieved = alg_bytes_launch / (avg_launch_ms * 1e-3) / 1e9
        # HBM traffic needs the PMC counters, i.e. a separate rocprofv3 --pmc run of this same command
        # (tools/profile_bench.sh): the line carries the committed figure of that run, and says so, when it was
        # taken for the same size / rank count / plan AND on the same kernel and planner sources; otherwise null
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
        if os.path.exists(pmc):
            try:
                p = json.load(open(pmc))
                if p.get("L") == L and p.get("n_gpus") == n_gpus and p.get("plan_signature") == plan_signature(mat):
                    if p.get("kernel_source_hash") == kernel_source_hash():
                        traffic = p.get("hbm_bytes_per_launch")
                        traffic_source = ("profiles/latest_pmc.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE of "
                                          "this command, separate run, same kernel sources (hash %s)" % p["kernel_source_hash"]
                                          + (", " + p["source"] if p.get("source") else ""))
                    else:
                        traffic_source = ("null: the kernel / planner sources have changed since the counter run of "
                                          "profiles/latest_pmc.json (tools/profile_bench.sh renews it)")
            except Exception:
                traffic, traffic_source = None, None
        two = launches == 2 and n_gpus == 1
        out = {
            "metric": "matrix-free H|psi> Gamplitudes/s, random-field Heisenberg",
            "value": dim / (ms_per_step * 1e-3) / 1e9,
            "unit": "Gamplitudes/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": f"L={L} random-field Heisenberg chain (benchmark.py 'MBL'), Full space, "
                                   f"2^{L} complex128 amplitudes, matrix-free y=Hx",
                       "L": L, "dim": dim, "nmasks": int(len(masks)), "nterms": int(H.msc.size),
                       "partition": f"{n_gpus} x 2^{L - int(math.log2(n_gpus))} contiguous blocks",
                       "baseline_config": BASELINE_CONFIG.get((n_gpus, L, args.model)),
                       "launches_per_step": launches,
                       "transport": (os.environ.get("DNM_BENCH_BACKEND", "nccl") if world > 1 else "none"),
                       "schedule": (None if world == 1 else
                                    "native (dnm_mat_mult_partitioned)" if native_default else "host (torch.distributed)"),
                       "amplitudes_per_gpu": dim_local,
                       "exchange_selfcheck": selfcheck,
                       **exchange_estimate(summary, None),
                       "plan": mat.describe().strip().replace("\n", " | "),
                       "plan_signature": plan_signature(mat)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         # what the memory system really sustained: counter bytes per launch / launch time / peak
                         "hbm_util": (traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "traffic_bytes_per_amp": (traffic * launches / dim_local) if traffic else None,
                         "floor_bytes_per_amp": FLOOR_BYTES_PER_AMP if launches == 2 else None,
                         "floor_derivation": FLOOR_DERIVATION if launches == 2 else None,
                         "frac_of_floor": (FLOOR_BYTES_PER_AMP * dim_local / launches / traffic) if (traffic and launches == 2) else None,
                         # the north star's target against what this algorithm class can reach: a sum over
                         # 29 bonds of a 30-cube takes two launches, 80 B/amp where 32 are credited -- 0.40 of the roofline
                         # at the 8 TB/s spec peak, 0.315 at the 6.3 TB/s this chip copies at
                         "target_frac": 0.60, "target_met": bool(achieved / HBM_PEAK_GBS >= 0.60),
                         "floor_frac_at_spec": (ALG_BYTES_PER_AMP / FLOOR_BYTES_PER_AMP) if two else None,
                         "copy_rate_GBs": COPY_RATE_GBS if two else None,
                         "floor_frac_at_copy_rate": (ALG_BYTES_PER_AMP / FLOOR_BYTES_PER_AMP * COPY_RATE_GBS / HBM_PEAK_GBS) if two else None,
                         "frac_vs_copy_ceiling": (FLOOR_BYTES_PER_AMP * dim_local / (kern_ms * 1e-3) / 1e9 / COPY_RATE_GBS) if two else None,
                         "target_note": ("0.60 of the 32 B/amp roofline is out of reach for complex128 at L=30 with any "
                                         "two-launch plan: the ceiling is floor_frac_at_spec; the kernel runs at "
                                         "frac_vs_copy_ceiling of what a copy of the same 80 B/amp would take") if two else None,
                         "kernel": "tile_pass_kernel", "avg_launch_ms": avg_launch_ms,
                         "alg_bytes_per_launch": alg_bytes_launch,
                         "read_only_frac": 0.5 * achieved / HBM_PEAK_GBS},
        }
        wd.partial = out

    if world > 1:
        # the multiply split three ways, for both schedules of the exchange
        mg = {"default_schedule": "native" if native_default else "host", "first_contact_probe": probe,
              "xgmi_link_GBs_assumed": XGMI_LINK_GBS, "schedules": {}}
        if out is not None:
            out["multi_gpu"] = mg
        nph = max(2, min(args.steps, 5))
        try:
            wd.phase("phases of the default schedule")
            w, e, c = phase_times(mat, x, y, nph, barrier, reduce_max)
            mg["schedules"]["native" if native_default else "host"] = schedule_entry(
                "native" if native_default else "host", w, e, c, summary, selfcheck)
            if out is not None:
                out["config"].update(exchange_estimate(summary, e * 1e-3))
        except Exception as e:       # noqa: BLE001
            mg["schedules"]["native" if native_default else "host"] = {"error": repr(e)}
        # ... and the other schedule: the host one after a native default; after a host default the native one only when
        # nothing has spoken against it (no failed probe) and the transport is RCCL (or a stand-in is named)
        other = "host" if native_default else "native"
        may = native_default or (probe is None and os.environ.get("DNM_NATIVE_COMM", "") != "0"
                                 and (dist.get_backend() == "nccl" or os.environ.get("DNM_RCCL_LIB")))
        if may:
            saved = config.native_comm
            try:
                wd.phase("%s schedule: build" % other)
                mat.destroy()
                mat = None
                torch.cuda.empty_cache()
                config.native_comm = (other == "native")
                mat = build()
                mat, verdict = warm(mat, 1, "%s schedule" % other)
                wd.phase("phases of the %s schedule" % other)
                w, e, c = phase_times(mat, x, y, nph, barrier, reduce_max)
                mg["schedules"][other] = schedule_entry(other, w, e, c, summary, verdict)
            except Exception as e:       # noqa: BLE001
                mg["schedules"][other] = {"error": repr(e)}
            finally:
                config.native_comm = saved
        else:
            mg["schedules"][other] = "not run: " + ("the first-contact probe failed" if probe is not None else
                                                    "DNM_NATIVE_COMM=0" if os.environ.get("DNM_NATIVE_COMM", "") == "0" else
                                                    "no RCCL transport (gloo-staged ranks)")

    if not args.no_secondary and ((n_gpus == 1 and L == 30) or n_gpus > 1):
        # the headline's vectors and operator go first: the solvers size their work space to the free memory
        if mat is not None:
            mat.destroy()
        mat = None
        del x, y
        torch.cuda.empty_cache()
        # (a failure in the Krylov phases -- memory for the work vectors, a failed check -- is recorded, it must not
        # cost the headline that has already been measured)
        sec = None
        try:
            if n_gpus == 1:
                sec = secondary(wd)
            else:
                Lk = tuple(int(v) for v in args.config5.split(",")) if args.config5 else None
                if Lk is None and dist.get_backend() != "nccl":
                    Lk = (26, 13)       # ranks sharing a device (a plumbing run): config 5's sizes are for a GPU per rank
                sec = {"config5": config5(wd, world, rank, Lk)}
        except Exception as e:       # noqa: BLE001
            sec = {"error": repr(e)}
        if out is not None:
            out["secondary"] = sec
            bad = [k for k, v in sec.items() if isinstance(v, dict) and ("failed_checks" in v or "error" in v or any(
                isinstance(w, dict) and "failed_checks" in w for w in v.values()))]
            out["secondary_ok"] = "error" not in sec and not bad
            if not out["secondary_ok"]:
                print("[bench] secondary phases with failed checks or errors: %s" % (bad or sec.get("error")), file=sys.stderr)
    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            wd.phase("cpu baseline")
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:       # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
        wd.phase("result line")
        wd.partial = None
        print(json.dumps(out))
        sys.stdout.flush()
    if mat is not None:
        mat.destroy()
    if world > 1:
        backend.release_native_comm()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
