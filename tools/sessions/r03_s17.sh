#!/bin/bash
# round 3, GPU session 17: lo pass launched per row-length class -- SpinConserve(32,16) multiply, config-5 rank, sc3 tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s17; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 900 python -m pytest tests/test_gpu_sc3.py -q -x 2>&1 | tail -5 | tee $OUT/pytest_sc3.txt
timeout 600 python tools/sc_bench.py 32 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_bench_32.txt
timeout 600 python tools/sc_bench.py 30 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_bench_30.txt
timeout 900 python tools/sc3_config5.py 2>&1 | grep -v amdgpu.ids | tee $OUT/config5.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_sc -o sc -- python3 $OLDPWD/tools/sc_bench.py 32 > /dev/null 2>&1; python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/prof_sc/**/*kernel_stats.csv', recursive=True):
    for i,r in enumerate(csv.reader(open(f))):
        if i<12: print(','.join(r[:6]))
PY
) | tee $OUT/sc32_kernel_stats.txt
