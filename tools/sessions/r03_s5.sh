#!/bin/bash
# round 3, GPU session 5: Chebyshev-filtered eigsolve(nev > 1); Krylov suite after the solver changes
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s5; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
{
echo "== filtered eigsolve"
DNM_KRYLOV_DEBUG=1 timeout 300 python tools/eigs_filter_bench.py 24 mbl 5 1e-10
DNM_KRYLOV_DEBUG=1 timeout 300 python tools/eigs_filter_bench.py 24 heisenberg 3 1e-8 highest
DNM_KRYLOV_DEBUG=1 timeout 600 python tools/eigs_filter_bench.py 26 xxz 5 1e-10
DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10
DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 30 mbl 3 1e-8 lowest --no-plain
} 2>&1 | tee $OUT/eigs_filter.txt
echo "== pytest krylov"; timeout 1200 python -m pytest tests/test_gpu_krylov.py -x -q 2>&1 | tail -8 | tee $OUT/pytest_krylov.txt
