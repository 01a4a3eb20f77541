#!/bin/bash
# round 5, GPU session 20b: the deflated recurrence as committed (projection at every step) on the tree with the partner
# table -- tests, kagome-30 forced through it, the 36-site torus (nev = 1, 2)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s20; mkdir -p $OUT
M=$OUT/deflation3.txt
timeout 900 python3 -m pytest tests/test_gpu_krylov.py tests/test_gpu_distributed.py -m gpu -q -x -k "deflated or fuzz or end_to_end" 2>&1 | tail -4 | tee $M
echo "== run_kagome 30, DNM_EIGS_BASISFREE=1 (deflation)" | tee -a $M
DNM_EXPERIMENTAL=1 DNM_EIGS_BASISFREE=1 DNM_KRYLOV_DEBUG=1 python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tail -8 | tee -a $M
echo "== kagome 36a, nev = 1 then nev = 2" | tee -a $M
DNM_TEST_LARGEST=1 DNM_KRYLOV_DEBUG=1 timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -s -k kagome36 2>&1 | grep -v amdgpu | tail -8 | tee -a $M
