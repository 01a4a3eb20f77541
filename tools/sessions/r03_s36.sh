#!/bin/bash
# round 3, GPU session 36: SpinConserve passes with table-driven LDS bonds -- parity, timing, instruction counts
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s36; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 900 python -m pytest tests/test_gpu_sc3.py -q -x 2>&1 | tail -4 | tee $OUT/pytest_sc3.txt
timeout 600 python tools/sc_bench.py 32 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_bench_32.txt
timeout 600 python tools/sc_bench.py 30 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_bench_30.txt
bash tools/prof_cmd.sh $OUT/sc32_kernel_stats.txt python3 tools/sc_bench.py 32 > /dev/null
bash tools/pmc_kernels.sh sc3_ 'SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES' -- python3 tools/sc_bench.py 32 | tee $OUT/sc32_valu.txt
