#!/bin/bash
# round 5, GPU session 22: real lo pass with consecutive entries per wavefront (one entry per lane instead of pairs):
# parity, timing, texture-path counters
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s22; mkdir -p $OUT
M=$OUT/stride1.txt
timeout 900 python3 -m pytest tests/test_gpu_sc3_graph.py tests/test_gpu_sc3.py -m gpu -q -x 2>&1 | tail -3 | tee $M
python3 tools/models_bench.py --real kagome30:sc kagome30:scx kagome27b:sc bench_long_range:sc:28 kagome33:sc 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "CASE\|multiply" | cut -c1-150 | tee -a $M
for G in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TD_TD_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES" "FETCH_SIZE"; do
  echo "-- kagome30:sc real: $G" | tee -a $M
  bash tools/pmc_kernels.sh sc3g_lo "$G" -- python3 tools/models_bench.py --real kagome30:sc 2>&1 | grep "pass_r\|rror" | tee -a $M
done
python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tail -3 | tee -a $M
