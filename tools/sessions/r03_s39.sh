#!/bin/bash
# round 3, GPU session 39: lo pass dispatched in pairs of rows (T, W), (T, W ^ 1) -- parity, timing, fetch counters
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s39; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 900 python -m pytest tests/test_gpu_sc3.py -q -x 2>&1 | tail -3 | tee $OUT/pytest_sc3.txt
timeout 600 python tools/sc_bench.py 32 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_bench_32.txt
bash tools/prof_cmd.sh $OUT/sc32_kernel_stats.txt python3 tools/sc_bench.py 32 > /dev/null
grep sc3_ $OUT/sc32_kernel_stats.txt | cut -c1-130
bash tools/pmc_kernels.sh sc3_ 'FETCH_SIZE' -- python3 tools/sc_bench.py 32 | tee $OUT/sc32_fetch.txt
timeout 900 python tools/sc3_config5.py --rank 3 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/config5_rank3.txt
