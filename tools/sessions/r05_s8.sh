#!/bin/bash
# round 5, GPU session 8: SQ counters of the bond-graph passes at kagome-30 (what they wait for), and the graph tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r05_s8; mkdir -p $OUT
M=$OUT/sq.txt
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  echo "-- $G" | tee -a $M
  bash tools/pmc_kernels.sh sc3g "$G" -- python3 tools/models_bench.py kagome30:sc | tee -a $M
done
timeout 600 python3 -m pytest tests/test_gpu_sc3_graph.py -q 2>&1 | tail -3 | tee -a $M
timeout 300 python3 tools/models_bench.py kagome30:sc kagome30:scx 2>&1 | grep "multiply" | cut -c1-100 | tee -a $M
