#!/bin/bash
# round 3, GPU session 8: internal SpinConserve layout incl. partitions, config-5 rank share, counters, filtered eigsolve
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s8; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
echo "== pytest sc3 + krylov(filtered) + distributed"
timeout 1500 python -m pytest tests/test_gpu_sc3.py tests/test_gpu_distributed.py -q 2>&1 | tail -25 | tee $OUT/pytest_sc3_dist.txt
timeout 900 python -m pytest tests/test_gpu_krylov.py -q -k "filtered or basis_free" 2>&1 | tail -15 | tee $OUT/pytest_krylov_filtered.txt
echo "== sc_bench and the prototype on the same box"
timeout 600 python tools/sc_bench.py 32 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_bench.txt
hipcc --offload-arch=gfx950 -O3 tools/experiments/sc3_proto.hip -o /tmp/sc3_proto && for v in 3 7; do timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 1 5 1024 1024 1 1 4 $v | tail -1; done 2>&1 | tee -a $OUT/sc_bench.txt
echo "== config 5, one rank's share"
for r in 3 2; do timeout 900 python tools/sc3_config5.py --rank $r 2>&1 | grep -v amdgpu.ids | tail -5; done | tee $OUT/sc3_config5.txt
{
echo "== PMC of the library's SpinConserve kernels, L=32 k=16"
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  echo "-- $G"
  timeout 600 tools/pmc_kernels.sh sc3_ "$G" -- python3 tools/sc_bench.py 32
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/sc3_pmc.txt
{
DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10 lowest
DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 26 xxz 5 1e-10 lowest --no-plain
} 2>&1 | grep -v amdgpu.ids | grep "^L=\|restarts,\|difference" | tee $OUT/eigs_filter.txt
