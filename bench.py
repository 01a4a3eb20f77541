#!/usr/bin/env python
"""
Headline benchmark: matrix-free H|psi> for the random-field Heisenberg chain
(reference benchmarking/benchmark.py 'MBL' Hamiltonian, `--shell --mult`),
Full space, complex128, on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--L L]

A "step" is one multiply y = H x with x, y resident in HBM.  N=1: L=30 (2^30
amplitudes, 16 GiB per vector; BASELINE.json configs[2]).  N>1: the state is
row-block partitioned, L = 30 + log2(N) so every GPU keeps 2^30 amplitudes
(weak scaling); partner blocks travel over RCCL while the rank-local masks
run.  Prints ONE JSON line on rank 0.
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


def cpu_baseline(sample_L=27, reps=3):
    """Oracle (C restatement of the reference's MatMult_CPU_Fast) timed on this
    box's host cores; a reported baseline, not the target."""
    import numpy as np
    from oracle import oracle as orc
    from dynamite_amd import models, msc_tools
    H = models.mbl(sample_L)
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    msc = orc.Msc(masks, offs, H.msc['signs'], H.msc['coeffs'])
    sub = orc.full(sample_L)
    n = 1 << sample_L
    rs = np.random.RandomState(0)
    x = rs.standard_normal(n) + 1j * rs.standard_normal(n)
    out = np.empty(n, dtype=np.complex128)
    nt = orc.max_threads()
    best = float('inf')
    for _ in range(reps):
        t0 = time.perf_counter()
        orc.matvec(msc, sub, sub, x, nthreads=nt, out=out)
        best = min(best, time.perf_counter() - t0)
    return {"value": n / best / 1e9, "unit": "Gamplitudes/s", "cores": nt, "kind": "port",
            "sample": f"L={sample_L} random-field Heisenberg, Full space, best of {reps} multiplies "
                      f"({best:.2f} s each) of the oracle's MatMult_CPU_Fast restatement, "
                      f"2^11-row blocks over {nt} OpenMP threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--model", default="mbl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus %d needs torch.distributed.run (one rank per GPU)" % args.gpus, file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        # rank % device count, as the reference picks its GPU (bcuda_template_2.cu:64-67)
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        # DNM_BENCH_BACKEND=gloo: dry run of the multi-rank flow with several ranks on one GPU (blocks staged
        # through the host; RCCL refuses two ranks on one device) -- a plumbing check, not a measurement
        backend_name = os.environ.get("DNM_BENCH_BACKEND", "nccl")
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend_name)
    from dynamite_amd import models, backend, _lib
    from dynamite_amd.config import config
    from dynamite_amd.subspaces import Full
    config._initialize()

    n_gpus = world
    L = args.L or (30 + int(math.log2(n_gpus)))
    H = models.BY_NAME[args.model](L)
    H.establish_L()
    H.reduce_msc()
    from dynamite_amd import msc_tools
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    sub = Full(L=L)
    mat = backend.build_mat(masks, offs, H.msc['signs'], H.msc['coeffs'], sub._to_c(), sub._to_c())
    if rank == 0:
        print(mat.describe(), file=sys.stderr)
    dim = 1 << L
    x, y = backend.Vec(dim), backend.Vec(dim)
    x.set_random(0)
    x.normalize()

    import ctypes as C
    nl = C.c_int()
    _lib.check(_lib.lib().dnm_mat_plan_launches(mat.handle, C.byref(nl)))
    launches = nl.value + len(mat.recvs)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        mat.mult(x, y)
    barrier()
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

    if rank == 0:
        # kernel time from HIP events on the launch stream (events bracket K steps
        # of back-to-back launches; per-launch = total / (K * launches per step))
        kern_ms = dev_ms / args.steps
        avg_launch_ms = kern_ms / launches
        dim_local = dim // n_gpus
        alg_bytes_launch = ALG_BYTES_PER_AMP * dim_local / launches
        achieved = alg_bytes_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
        if os.path.exists(pmc):
            try:
                p = json.load(open(pmc))
                if p.get("L") == L and p.get("n_gpus") == n_gpus and p.get("plan") == os.environ.get("DNM_PLAN_MODE", "2"):
                    traffic = p.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
                       "launches_per_step": launches,
                       "tile_bits": int(os.environ.get("DNM_TILE_BITS", "12")),
                       "plan_mode": int(os.environ.get("DNM_PLAN_MODE", "2")),
                       "plan": mat.describe().strip().replace("\n", " | ")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "tile_pass_kernel", "avg_launch_ms": avg_launch_ms,
                         "alg_bytes_per_launch": alg_bytes_launch,
                         "read_only_frac": 0.5 * achieved / HBM_PEAK_GBS},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    mat.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
