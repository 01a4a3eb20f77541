#!/bin/bash
# round 5, GPU session 21: what the bond-graph lo pass waits for after the partner table: texture / L1 / L2 counters
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s21; mkdir -p $OUT
M=$OUT/lo_pass_memory_counters.txt
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TA_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*sum\)\b" | sort -u | tr '\n' ' ' | cut -c1-6000 > $OUT/avail.txt
for G in "TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_TD_BUSY_sum TD_BUSY_avr"; do
  echo "-- kagome30:sc real: $G" | tee -a $M
  bash tools/pmc_kernels.sh sc3g "$G" -- python3 tools/models_bench.py --real kagome30:sc 2>&1 | grep "pass_r\|false, true>\|rror" | tee -a $M
done
