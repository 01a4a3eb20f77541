#!/bin/bash
# round 5, GPU session 37: window-pass column block (DNM_SC3G_WBLOCK) and lo-pass order for the REAL bond-graph multiply
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s37; mkdir -p $OUT
M=$OUT/real_knobs.txt
for B in 4 5 6 7 8; do
  echo "== DNM_SC3G_WBLOCK=$B" | tee -a $M
  DNM_SC3G_WBLOCK=$B python3 tools/models_bench.py --real kagome30:sc kagome30:scx 2>&1 | grep "real arith\|  multiply " | cut -c1-110 | tee -a $M
done
for O in 0 1 2; do
  echo "== DNM_SC3G_ORDER=$O" | tee -a $M
  DNM_SC3G_ORDER=$O python3 tools/models_bench.py --real kagome30:sc 2>&1 | grep "real arith\|  multiply " | cut -c1-110 | tee -a $M
done
