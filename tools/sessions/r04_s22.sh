#!/bin/bash
# round 4, GPU session 22: where the time of the two tiled passes goes, by SQ / TA / TCP counters (closing record)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s22; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_THREAD_CYCLES_VALU SQ_CYCLES" \
         "GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  echo "-- $G" | tee -a $OUT/sq_counters.txt
  NLAST=2 bash tools/pmc_kernels.sh tile_pass "$G" -- $CMD | tee -a $OUT/sq_counters.txt
done
