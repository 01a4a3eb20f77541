#!/bin/bash
# round 3, GPU session 30: vector sweeps with one element per thread / more reduction workgroups -- ABI timings, solver timings, tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s30; mkdir -p $OUT
timeout 600 python tools/vec_abi_bench.py 30 2>&1 | grep -v amdgpu.ids | tee $OUT/vec_abi_30.txt
timeout 600 python tools/vec_abi_bench.py 24 2>&1 | grep -v amdgpu.ids | tee $OUT/vec_abi_24.txt
timeout 900 python tools/krylov_L30.py 30 2>&1 | grep -v amdgpu.ids | tee $OUT/krylov_L30.txt
timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10 2>&1 | grep -v amdgpu.ids | tee $OUT/eigs_filter_28.txt
timeout 900 python tools/sc_eigs_bench.py 2>&1 | grep -v amdgpu.ids | head -4 | tee $OUT/sc_eigs.txt
timeout 2400 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_vec.py -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
