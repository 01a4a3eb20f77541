#!/bin/bash
# round 3, GPU session 12: partitioned runs after the local / remote split of the internal-layout multiply
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s12; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 2400 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_sc3.py tests/test_gpu_fullsize.py -q 2>&1 | tail -15 | tee $OUT/pytest.txt
