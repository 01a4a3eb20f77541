#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp35_probe.txt
echo "# evolve at memory-bound sizes: Lanczos probe of the norm bound instead of acquiring the Krylov workspace" > $O
timeout 1500 python3 -m pytest tests/test_gpu_krylov.py tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -4 >> $O
DNM_KRYLOV_DEBUG=1 timeout 900 python3 tools/krylov_L30.py 2>&1 | grep -v amdgpu.ids >> $O
timeout 900 python3 tools/cheb_bench.py 30 2>&1 | grep -v amdgpu.ids | tail -12 >> $O
