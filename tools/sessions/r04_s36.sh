#!/bin/bash
# round 4, GPU session 36: counters of rdm_mfma_kernel (L=26, keep = 13 low spins)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r04_s36; mkdir -p $OUT
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM"; do
  echo "-- $G" | tee -a $OUT/pmc.txt
  bash tools/pmc_kernels.sh rdm_mfma "$G" -- python3 tools/rdm_bench.py 26 13 | tee -a $OUT/pmc.txt
done
