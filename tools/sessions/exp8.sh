#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp8_fullsize.txt
echo "# full per-rank-size tests (configs 4, 5) + bench with the new cpu baseline" > $O
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --durations=5 2>&1 | tail -25 >> $O
timeout 900 python3 bench.py --steps 10 --warmup 2 2>/dev/null >> $O
