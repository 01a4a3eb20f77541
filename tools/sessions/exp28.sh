#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp28_evolve.txt
echo "# evolve: expansion chosen outright when it costs less than one Krylov outer step" > $O
timeout 1500 python3 -m pytest tests/test_gpu_krylov.py -x -q -m gpu 2>&1 | tail -4 >> $O
timeout 900 python3 tools/krylov_L30.py 2>&1 | grep -v amdgpu >> $O
timeout 600 python3 tools/cheb_bench.py 26 2>&1 | grep -v amdgpu | tail -8 >> $O
