#!/bin/bash
# round 3, GPU session 23: tile kernel with scalar-base + 32-bit-offset addressing -- parity, timing
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s23; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 1500 python -m pytest tests/test_gpu_matvec.py -q -x 2>&1 | tail -5 | tee $OUT/pytest_matvec.txt
timeout 600 python tools/policy_sizes.py 20 22 24 26 28 30 2>&1 | grep -v amdgpu.ids | grep "policy 226" | tee $OUT/sizes.txt
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/bench.json
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_krylov.py -q -x 2>&1 | tail -5 | tee $OUT/pytest_rest.txt
