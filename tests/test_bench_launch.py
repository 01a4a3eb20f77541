"""
`python bench.py --gpus N` without a launcher must start its own N rank processes
(never re-exec a process that has touched the GPU) and print ONE JSON line from
rank 0.  Without a GPU the ranks run the dry-run flow: every rank's plan, the
exchange schedule (bpetsc_template_2.c:787-879 / bcuda_template_2.cu:161-171 in the
reference) and the gloo transport, with the xGMI volume the line reports.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 4])
def test_bench_spawns_its_ranks(n):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("dry-run flow is the no-GPU path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2",
                        "--warmup", "1", "--L", "16"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["dry_run"] is True and out["exchange_ok"] is True
    assert out["value"] is None and out["scaling"] == "weak"
    # every key of the N > 1 line is there (what cannot run without a GPU says so): both schedules of the exchange, each
    # split three ways, the probe's verdict, config 5
    mg = out["multi_gpu"]
    assert set(mg) == {"default_schedule", "first_contact_probe", "xgmi_link_GBs_assumed", "schedules"}
    assert set(mg["schedules"]) == {"host", "native"} and mg["schedules"]["native"].startswith("not run")
    host = mg["schedules"]["host"]
    assert set(host) == {"schedule", "ms_per_step", "exchange_only_ms", "compute_only_ms", "hidden_ms",
                         "hidden_frac_of_the_shorter", "busiest_link_bytes", "link_GBs_measured", "selfcheck"}
    assert host["exchange_only_ms"] > 0 and host["link_GBs_measured"] > 0 and host["compute_only_ms"] is None
    c5 = out["secondary"]["config5"]
    assert c5["exchange"] == "window" and c5["exchange_ok"] is True and c5["dim"] == 48620
    assert 0 < c5["bytes_received_per_multiply_busiest_rank"] < 16 * c5["dim"]
    # ... on the partition made for the exchange: less than the reference-compatible one moves
    assert c5["bytes_received_per_multiply_busiest_rank"] <= c5["bytes_received_per_multiply_busiest_rank_reference_compatible_partition"]
    assert {"heisenberg", "known_answer_xx_chain"} <= set(c5) and out["secondary_ok"] is True
    assert out["config"]["schedule"].startswith("host") and "exchange_selfcheck" in out["config"]
    cfg = out["config"]
    nloc = (1 << 16) // n
    if n < 4:
        # partner blocks: rank 0 (all rank bits 0) receives half a block across the block boundary and nothing for
        # the bonds inside the rank bits (its rows are annihilated there): 2^(L - log2 n - 1) amplitudes of 16 B
        assert cfg["exchange"] == "partner"
        assert cfg["xgmi_bytes_per_step"] == 16 * (nloc // 2)
        assert cfg["xgmi_partners"] == 1 and cfg["xgmi_link_bound_ms"] > 0
    else:
        # transposed exchange: (n-1)/n of the block goes out and comes back, the same share on every link
        assert cfg["exchange"] == "transpose" and cfg["plan"].startswith("transposed exchange")
        assert cfg["xgmi_bytes_per_step"] == 2 * 16 * nloc * (n - 1) // n
        assert cfg["xgmi_partners"] == n - 1 and cfg["xgmi_busiest_link_bytes"] == 2 * 16 * nloc // n


def _run_bench(argv, extra_env=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                          timeout=timeout, env=env)


def test_bench_eight_ranks_dry_run():
    """The 8-rank launch the driver's SCALE run uses, at a size eight CPU processes can hold: transposed exchange,
    every link carrying 2 * block / 8 there and back, and the link rate measured in the run next to the assumed."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("dry-run flow is the no-GPU path")
    p = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--L", "23"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    cfg = out["config"]
    nloc = (1 << 23) // 8
    assert out["n_gpus"] == 8 and out["exchange_ok"] is True
    assert cfg["exchange"] == "transpose" and cfg["xgmi_partners"] == 7
    assert cfg["xgmi_busiest_link_bytes"] == 2 * 16 * nloc // 8
    assert cfg["xgmi_link_GBs_assumed"] == 64.0
    assert cfg["xgmi_exchange_only_ms"] > 0 and cfg["xgmi_link_GBs_measured"] > 0


def test_bench_default_sizes():
    """N=1 is BASELINE configs[2] (L=30), N=8 is configs[3] (L=34); 2 and 4 keep 2^30 amplitudes per GPU."""
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.default_L(n) for n in (1, 2, 4, 8)] == [30, 31, 32, 34]
    assert "configs[3]" in bench.BASELINE_CONFIG[(8, 34, "mbl")]
    assert "configs[2]" in bench.BASELINE_CONFIG[(1, 30, "mbl")]


@pytest.mark.parametrize("rank", [0, 5])
def test_config4_plan_on_the_host(rank):
    """BASELINE configs[3] -- L=34 on 8 ranks, 2^31 amplitudes per rank -- planned with a host-only handle: the
    transposed exchange applies, both of its operators are rank-local, the layout-A operator runs the two-launch
    plan on n_loc = 31 and every link carries 2 * 2^31 / 8 amplitudes of 16 B per multiply."""
    from dynamite_amd import models, backend, msc_tools, _lib
    from dynamite_amd.subspaces import Full
    import bench
    L, world = 34, 8
    H = models.mbl(L)
    H.establish_L()
    H.reduce_msc()
    masks, offs = msc_tools.get_mask_offsets(H.msc)
    assert len(masks) == 34
    sub = Full(L=L)
    lc, rc = sub._c(), sub._c()
    # the layout the 8-rank run takes: the swizzle field must end below the pieces of the all-to-all
    lc.vec_swizzle = rc.vec_swizzle = 14
    split = backend.transpose_split(masks, offs, H.msc['signs'], H.msc['coeffs'], L, world, int(lc.vec_swizzle))
    assert split is not None
    lo, hi, f = split
    assert f == 31 - 1 - 3
    import ctypes as C
    descr = []
    for which, part in enumerate((lo, hi)):
        # layout B as ShellMat.set_transposed builds it: tiles [0, 8) + [f, n) so that ranges of workgroups are
        # contiguous sub-pieces of the all-to-all's pieces
        flags = _lib.MAT_HOST_ONLY | ((12 - (31 - f)) << _lib.MAT_AMIN_SHIFT if which == 1 else 0)
        h = backend.create_mat(*part, lc, rc, False, flags, rank, world)
        assert backend.exchange_plan(h) == ([], [])
        descr.append(bench.C_describe(h))
        if which == 1:
            top, gathers = C.c_int(), C.c_int()
            _lib.check(_lib.lib().dnm_mat_local_part_bits(h, C.byref(top), C.byref(gathers)))
            assert (top.value, gathers.value) == (f - 1, 0)
        _lib.check(_lib.lib().dnm_mat_destroy(h))
    assert "n=34 n_loc=31" in descr[0] and "tiled=1" in descr[0] and descr[0].count("local pass") == 2
    assert "tiled=1" in descr[1] and descr[1].count("local pass") == 1        # layout B: one LDS-only window launch
    assert "segs [0,8) [27,31)" in descr[1] and "gather_masks=0" in descr[1]
    pieces, own, cnt = backend.transpose_pieces(31, 3, f, rank)
    per_peer = {}
    for q, _, c in pieces:
        per_peer[q] = per_peer.get(q, 0) + 16 * c
    assert len(per_peer) == 7 and set(per_peer.values()) == {16 * (1 << 31) // 8}


def test_bench_watchdog_ends_a_hung_run():
    """One rank never posts its exchange: the others' watchdogs give up with the phase and the plan printed, the
    parent ends what is left and returns non-zero -- nothing waits forever."""
    import time
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("dry-run flow is the no-GPU path")
    t0 = time.time()
    p = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--L", "14", "--watchdog", "4"],
                   {"DNM_BENCH_TEST_HANG_RANK": "1"}, timeout=300)
    assert p.returncode != 0
    assert time.time() - t0 < 120
    assert "[bench watchdog] rank" in p.stderr and "no progress in phase" in p.stderr and "plan:" in p.stderr
    assert "ending the other ranks" in p.stderr


def test_line_keys_are_the_same_with_and_without_a_gpu():
    """The keys the GPU run's `multi_gpu` entries and config 5 carry are those of the dry run (bench.schedule_entry is the
    one constructor of both); config 5's sizes by rank count keep about 1.1 G rows per GPU, 8 ranks = BASELINE configs[4]."""
    import math
    sys.path.insert(0, ROOT)
    import bench
    e = bench.schedule_entry("native", 10.0, 8.0, 6.0, {"busiest_link_bytes": 1 << 30}, "ok")
    assert e["hidden_ms"] == 4.0 and abs(e["hidden_frac_of_the_shorter"] - 4.0 / 6.0) < 1e-12
    assert abs(e["link_GBs_measured"] - (1 << 30) / 8e-3 / 1e9) < 1e-9
    assert bench.CONFIG5_BY_WORLD[8] == (36, 18)
    rows = {w: math.comb(*lk) / w for w, lk in bench.CONFIG5_BY_WORLD.items()}
    assert all(1.1e9 < r < 1.2e9 for r in rows.values()), rows


def test_probe_child_environment():
    """The probe's children make their own rendezvous: another port, no torch-elastic agent store, the schedule forced."""
    sys.path.insert(0, ROOT)
    import bench
    import inspect
    src = inspect.getsource(bench.first_contact_probe)
    assert "TORCHELASTIC_" in src and "PROBE_PORT_OFFSET" in src and 'env["DNM_NATIVE_COMM"]' in src
    assert "p.kill()" in src and "pkill" not in src          # the exact child, never a pattern
