#!/bin/bash
# round 3, GPU session 31: vector sweeps, reduction grid scaled with the size -- ABI timings at three sizes, Krylov tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s31; mkdir -p $OUT
for L in 30 26 22; do timeout 600 python tools/vec_abi_bench.py $L 2>&1 | grep -v amdgpu.ids | tee $OUT/vec_abi_$L.txt; done
timeout 3000 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_sc3.py -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
