#!/bin/bash
# round 5, GPU session 38: the fuzz tests with many more seeds than the suite runs (a bug hunt, not a gate)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s38; mkdir -p $OUT
export DNM_FUZZ_N=400 DNM_FUZZ_REAL_N=200 DNM_FUZZ_XPARITY_N=150 DNM_SC3_FUZZ_N=300 DNM_SC3_FUZZ_REAL_N=200 DNM_SC3G_FUZZ_N=300 DNM_SC3G_FUZZ_X_N=100 DNM_FUZZ_EIGS_REAL_N=40 DNM_FUZZ_KRYLOV_N=24 DNM_FUZZ_RDM_N=100
timeout 2400 python3 -m pytest tests/test_gpu_matvec.py tests/test_gpu_sc3.py tests/test_gpu_sc3_graph.py tests/test_gpu_krylov.py -m gpu -q -k "fuzz" 2>&1 | tail -30 | cut -c1-250 | tee $OUT/fuzz_wide.txt
