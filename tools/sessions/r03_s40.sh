#!/bin/bash
# round 3, GPU session 40: counters of the final SpinConserve kernels (SpinConserve(32,16))
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s40; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
{
echo "== PMC of the library's SpinConserve kernels at the end of round 3, L=32 k=16 (separate passes per counter group)"
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  echo "-- $G"
  bash tools/pmc_kernels.sh sc3_ "$G" -- python3 tools/sc_bench.py 32 | grep -E "sc3_|void dnm|dnm::" | grep -v "random\|copy"
done
python3 tools/sc_bench.py 32 2>&1 | grep -v amdgpu.ids
} | tee $OUT/sc3_pmc_final.txt
