#!/bin/bash
# round 5, GPU session 42: the lo pass's LDS hops by rank tables (no partner table): parity
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s42; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_sc3_graph.py -m gpu -q -k "rank_tables" --durations=6 2>&1 | tail -30 | cut -c1-220 | tee $OUT/rank_tables.txt
