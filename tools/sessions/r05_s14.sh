#!/bin/bash
# round 5, GPU session 14: the real-arithmetic bond-graph passes -- kernel times and SQ counters
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s14; mkdir -p $OUT
M=$OUT/kagome_real.txt
python3 tools/models_bench.py --real kagome30:sc kagome30:scx 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "CASE\|multiply" | cut -c1-200 | tee $M
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "FETCH_SIZE"; do
  echo "-- kagome30:sc real: $G" | tee -a $M
  bash tools/pmc_kernels.sh sc3g "$G" -- python3 tools/models_bench.py --real kagome30:sc | grep "pass_r\|true, true>\|false, true>" | tee -a $M
done
