#!/bin/bash
# round 3, GPU session 6: SpinConserve internal layout in the library (tests, timing, counters), filtered eigsolve v2
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s6; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
echo "== pytest sc3"; timeout 1200 python -m pytest tests/test_gpu_sc3.py -x -q 2>&1 | tail -30 | tee $OUT/pytest_sc3.txt
echo "== sc_bench"; timeout 600 python tools/sc_bench.py 28 32 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_bench.txt
{
echo "== filtered eigsolve v2"
DNM_KRYLOV_DEBUG=1 timeout 600 python tools/eigs_filter_bench.py 26 xxz 5 1e-10 lowest --no-plain
DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10 lowest --no-plain
DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 30 mbl 3 1e-8 lowest --no-plain
} 2>&1 | grep -v amdgpu.ids | tee $OUT/eigs_filter.txt
echo "== pytest -m gpu (all)"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
