#!/bin/bash
# round 3, GPU session 15: pipelined transposed exchange -- staged multi-rank tests, timeline
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s15; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 2400 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -q 2>&1 | tail -15 | tee $OUT/pytest.txt
timeout 900 python tools/transpose_timeline.py 27 4 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tee $OUT/transpose_timeline.txt
timeout 900 python tools/transpose_timeline.py 28 8 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tee -a $OUT/transpose_timeline.txt
DNM_TRANSPOSE_PIPE=0 timeout 900 python tools/transpose_timeline.py 28 8 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tee -a $OUT/transpose_timeline.txt
