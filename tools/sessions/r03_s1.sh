#!/bin/bash
# round 3, GPU session 1: suite after the pruning, counters of today's SpinConserve block kernel, copy-rate probe
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s1; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== sc_bench 32"; timeout 300 python tools/sc_bench.py 32 2>&1 | tee $OUT/sc_bench_32.txt
echo "== sc_block PMC" | tee $OUT/sc_block_pmc.txt
for G in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  echo "-- $G" | tee -a $OUT/sc_block_pmc.txt
  timeout 300 tools/pmc_cmd.sh sc_block_kernel "$G" -- python3 tools/sc_bench.py 32 2>&1 | tee -a $OUT/sc_block_pmc.txt
done
echo "== copy probe"
hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o /tmp/copy_probe && timeout 300 /tmp/copy_probe 2>&1 | tee $OUT/copy_probe.txt
