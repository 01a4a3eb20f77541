#!/bin/bash
# round 4, GPU session 42: validation of the tree as committed -- full GPU suite, smoke, default bench line, kernel stats of the same command
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s42; mkdir -p $OUT
timeout 2700 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 600 python3 bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_line.json; cut -c1-300 $OUT/bench_line.json
rm -rf /tmp/rp_final
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp_final -o t -- python3 bench.py --no-cpu-baseline > $OUT/line_under_rocprof.json 2> $OUT/rocprof.err
find /tmp/rp_final -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -8 $OUT/kernel_stats.csv | cut -c1-160
