#!/bin/bash
# round 5, GPU session 19: where eigsolve(nev=2) of kagome-30 spends its time (kernel totals)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s19; mkdir -p $OUT
for c in kagome30:sc kagome30:scx; do
  bash tools/prof_cmd.sh $OUT/eigs_$c.txt python3 tools/models_bench.py --eigs $c
  head -24 $OUT/eigs_$c.txt | cut -c1-200
done
