#!/bin/bash
# round 5, GPU session 26: real arithmetic under XParity on the Full space -- tests, then timing at L=28 / 30
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s26; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_krylov.py -m gpu -q -x -k "real_arithmetic" 2>&1 | tail -12 | tee $OUT/xparity_full_real.txt
python3 tools/models_bench.py --real --eigs ising:fullx:28 heisenberg:fullx:30 2>&1 | grep -v "Warning\|amdgpu.ids\|plan:" | cut -c1-200 | tee -a $OUT/xparity_full_real.txt
