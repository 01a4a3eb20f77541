#!/bin/bash
# round 4, GPU session 23: the scalar diet of the tiled kernel (host-computed k positions and position-space gather
# masks, short block deposit) against the previous kernel on the same box, alternating; parity first
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s23; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_matvec.py tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/parity.txt
BASE=$PWD/dynamite_amd/build/exp/lib_base.so
for i in 1 2 3; do
  bash tools/pass_times.sh base$i DNM_LIB=$BASE | tee -a $OUT/ab.txt
  bash tools/pass_times.sh new$i | tee -a $OUT/ab.txt
done
for i in 1 2; do
  DNM_EXPERIMENTAL=1 DNM_LIB=$BASE python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-200 | tee -a $OUT/ab.txt
  python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-200 | tee -a $OUT/ab.txt
done
# SQ instruction counts of the new kernel (one group)
DNM_EXPERIMENTAL=1 NLAST=2 bash tools/pmc_kernels.sh tile_pass "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary | tee $OUT/sq_new.txt
