#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp34_window_ranges.txt
echo "# window partitions ship only the ranges a rank reads" > $O
timeout 1500 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15 >> $O
