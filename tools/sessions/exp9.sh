#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp9_window_partitions.txt
echo "# window partitions for Explicit / Auto / projections / odd rank counts" > $O
timeout 2400 python3 -m pytest tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -40 >> $O
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8 >> $O
