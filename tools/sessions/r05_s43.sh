#!/bin/bash
# round 5, GPU session 43: dnm_comm_prepare in the ctypes-only child; rank-table tests as trimmed
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s43; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_sc3_graph.py -m gpu -q -k "native or rank_tables" 2>&1 | tail -12 | cut -c1-220 | tee $OUT/native_prepare.txt
