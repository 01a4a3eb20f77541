#!/bin/bash
# round 3, GPU session 24: gather records two at a time -- parity, timing
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s24; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 1500 python -m pytest tests/test_gpu_matvec.py -q -x 2>&1 | tail -3 | tee $OUT/pytest_matvec.txt
timeout 600 python tools/policy_sizes.py 20 22 24 26 28 30 2>&1 | grep -v amdgpu.ids | grep "policy 226" | tee $OUT/sizes.txt
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | tee $OUT/bench.json
