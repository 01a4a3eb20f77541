#!/bin/bash
# round 3, GPU session 17b: per-kernel times of the class-split lo pass
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s17; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
rocprofv3 --kernel-trace --stats -d /tmp/prof_sc -o sc -- python3 tools/sc_bench.py 32 > /dev/null 2>&1
python3 - <<'PY' | tee $OUT/sc32_kernel_stats.txt
import csv,glob
for f in glob.glob('/tmp/prof_sc/**/*kernel_stats.csv', recursive=True):
    for i,r in enumerate(csv.reader(open(f))):
        if i<12: print(','.join(x[:150] for x in r[:6]))
PY
