#!/bin/bash
# round 5, GPU session 15: partner table for the lo pass's LDS hops (Sc3Op::ptab) -- parity, then A/B against the rank tables
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s15; mkdir -p $OUT
M=$OUT/ptab.txt
timeout 900 python3 -m pytest tests/test_gpu_sc3_graph.py tests/test_gpu_sc3.py -m gpu -q -x 2>&1 | tail -6 | tee $M
for V in 0 1; do
  echo "== DNM_SC3G_PTAB=$V" | tee -a $M
  DNM_SC3G_PTAB=$V python3 tools/models_bench.py --real kagome30:sc kagome30:scx kagome27b:sc bench_long_range:sc:28 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "CASE\|multiply" | cut -c1-150 | tee -a $M
done
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  echo "-- kagome30:sc: $G" | tee -a $M
  bash tools/pmc_kernels.sh sc3g_lo "$G" -- python3 tools/models_bench.py --real kagome30:sc | tee -a $M
done
python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tail -3 | tee -a $M
