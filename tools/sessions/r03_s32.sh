#!/bin/bash
# round 3, GPU session 32: copy kernel in place of device-to-device memcpy -- timing, distributed + Krylov tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s32; mkdir -p $OUT
timeout 600 python tools/vec_abi_bench.py 30 2>&1 | grep -v amdgpu.ids | tee $OUT/vec_abi_30.txt
timeout 3000 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_distributed.py -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
