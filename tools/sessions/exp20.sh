#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp20_basisfree.txt
echo "# basis-free Lanczos for one extremal pair" > $O
timeout 900 python3 -m pytest tests/test_gpu_krylov.py -x -q -m gpu -k "basis_free or eigsolve" 2>&1 | tail -8 >> $O
timeout 900 python3 -m pytest tests/test_gpu_distributed.py -x -q -m gpu -k "full-2 or sc-3" 2>&1 | tail -4 >> $O
export DNM_KRYLOV_DEBUG=1
echo "== L=30 default (basis-free)" >> $O
timeout 900 python3 tools/krylov_L30.py 2>&1 | grep -v amdgpu >> $O
echo "== L=30 restarted (DNM_EIGS_BASISFREE=0)" >> $O
DNM_EIGS_BASISFREE=0 timeout 900 python3 tools/krylov_L30.py 2>&1 | grep "eigsolve" >> $O
echo "== SpinConserve(32,16)" >> $O
timeout 900 python3 tools/sc_eigs_bench.py 32 1e-8 2 2>&1 | grep -v amdgpu >> $O
DNM_EIGS_BASISFREE=0 timeout 900 python3 tools/sc_eigs_bench.py 32 1e-8 2 2>&1 | grep -v amdgpu >> $O
