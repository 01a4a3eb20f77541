#!/bin/bash
# round 3, GPU session 33: where the solvers' device time goes at L=30 / L=28 (rocprofv3 kernel stats)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s33; mkdir -p $OUT
bash tools/prof_cmd.sh $OUT/krylov_L30_stats.txt python3 tools/krylov_L30.py 30 > /dev/null
bash tools/prof_cmd.sh $OUT/eigs_filter_28_stats.txt python3 tools/eigs_filter_bench.py 28 mbl 5 1e-10 lowest --no-plain > /dev/null
