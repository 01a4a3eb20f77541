#!/bin/bash
# round 5, GPU session 35: bench line with the known-answer entry; entropy test with the Renyi check
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s35; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "half_chain_entropy" 2>&1 | tail -3 | tee $OUT/entropy.txt
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_s35/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k, v in d["secondary"].items():
    print(k, {a: b for a, b in v.items() if a in ("wall_s", "E0", "exact", "abs_error", "failed_checks", "E0_per_site")} if isinstance(v, dict) else v)
PY
