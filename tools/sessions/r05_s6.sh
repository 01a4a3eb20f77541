#!/bin/bash
# round 5, GPU session 6: the bound for VERDICT task 3 (top T bonds dropped from the chain), the partitioned bond-graph
# case, kagome-33 eigsolve after the sizing fix
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s6; mkdir -p $OUT
timeout 600 python3 tools/sc3_drop_bond_probe.py 32 2>&1 | grep -v amdgpu | tee $OUT/drop_bond.txt
for d in 0 1 2; do
  echo "-- per-pass times, top $d bond(s) dropped" | tee -a $OUT/drop_bond.txt
  DROP=$d bash tools/prof_cmd.sh $OUT/prof_drop$d.txt python3 tools/sc3_drop_bond_probe.py 32 2>&1 | grep "sc3_win\|sc3_lo" | head -4 | tee -a $OUT/drop_bond.txt
done
timeout 900 python3 -m pytest tests/test_gpu_distributed.py -q -k "sc3" 2>&1 | tail -5 | tee $OUT/tests.txt
timeout 900 python3 tools/models_bench.py --eigs kagome33:sc 2>&1 | grep "CASE\|multiply\|eigsolve\|Error" | cut -c1-200 | tee $OUT/kagome33.txt
