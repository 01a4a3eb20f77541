#!/bin/bash
# round 5, GPU session 12: the transposed exchange split and scheduled natively (loop-back against the oracle, timing at
# full size), the deflated eigensolver's remaining tests, kagome-30 by deflation against the filtered restart
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s12; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_distributed.py -m gpu -q -x -k "native or rccl" 2>&1 | tail -15 | tee $OUT/native_transposed.txt
timeout 900 python3 -m pytest tests/test_gpu_krylov.py -m gpu -q -x -k "deflated" 2>&1 | tail -5 | tee -a $OUT/native_transposed.txt
echo "== run_kagome 30, DNM_EIGS_BASISFREE=1 (deflation)" | tee -a $OUT/native_transposed.txt
DNM_EXPERIMENTAL=1 DNM_EIGS_BASISFREE=1 DNM_KRYLOV_DEBUG=1 python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tail -8 | tee -a $OUT/native_transposed.txt
timeout 1200 python3 tools/rccl_loopback_bench.py 2>&1 | grep -v "amdgpu\|Warning" | tee $OUT/rccl_loopback.txt
