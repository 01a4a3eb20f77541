#!/bin/bash
# round 3, GPU session 35: filter normalisation fused into the Lanczos sweep -- eigsolve tests, L=28 nev=5 timing
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s35; mkdir -p $OUT
timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10 lowest --no-plain 2>&1 | grep -v amdgpu.ids | tee $OUT/eigs_filter_28.txt
timeout 900 python tools/eigs_filter_bench.py 26 xxz 5 1e-10 lowest --no-plain 2>&1 | grep -v amdgpu.ids | tee $OUT/eigs_filter_26.txt
timeout 2400 python -m pytest tests/test_gpu_krylov.py -q -x -k "eigsolve" 2>&1 | tail -4 | tee $OUT/pytest.txt
