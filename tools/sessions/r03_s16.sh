#!/bin/bash
# round 3, GPU session 16: one rank's compute share of the transposed exchange at full size, new layout-B tile; staged tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s16; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 900 python tools/transpose_probe.py 30 8 5 2>&1 | grep -v amdgpu.ids | tee $OUT/transpose_probe.txt
timeout 900 python tools/transpose_probe.py 30 4 2 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $OUT/transpose_probe.txt
timeout 1500 python -m pytest tests/test_gpu_distributed.py -q 2>&1 | tail -5 | tee $OUT/pytest.txt
