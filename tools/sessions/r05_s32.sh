#!/bin/bash
# round 5, GPU session 32: half-chain entropy of the filled Fermi sea (Peschel) through eigsolve + the RDM kernels; then the
# full suite with the BASELINE-size known-answer tests in it
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s32; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "half_chain_entropy" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-220 | tee $OUT/entropy.txt
timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 2>&1 | tail -25 > $OUT/test_durations.txt; tail -3 $OUT/test_durations.txt
