#!/bin/bash
# round 5, GPU session 40: chains through the bond-graph passes again (DNM_SC3_GRAPH=1) now that their LDS hops come by table
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s40; mkdir -p $OUT
M=$OUT/chain_through_graph.txt
for G in 0 1 0 1; do
  echo "== DNM_SC3_GRAPH=$G" | tee -a $M
  DNM_SC3_GRAPH=$G python3 tools/models_bench.py --real heisenberg:sc:32 mbl:sc:30 2>&1 | grep "CASE\|multiply\|plan" | cut -c1-170 | tee -a $M
done
