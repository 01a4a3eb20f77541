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
