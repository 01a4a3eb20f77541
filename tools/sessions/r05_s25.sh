#!/bin/bash
# round 5, GPU session 25: the other subspaces on the tiled kernel -- XParity on the Full space, Parity -- for the table
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s25; mkdir -p $OUT
python3 tools/models_bench.py --real ising:fullx:28 heisenberg:fullx:28 heisenberg:parity:28 ising:full:28 bench_long_range:fullx:28 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-230 | tee $OUT/other_subspaces.txt
