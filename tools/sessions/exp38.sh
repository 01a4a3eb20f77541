#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp38b_rows4.txt
echo "# 4 rows per thread at 8 waves per SIMD against the default 8 rows at 4 waves, alternating, same box" > $O
one() { timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2 3 4 5 6; do
  echo "rows=4" >> $O; DNM_LOG_ROWS=2 one >> $O
  echo "rows=8" >> $O; one >> $O
done
bash tools/pass_times.sh r4 DNM_LOG_ROWS=2 >> $O
bash tools/pass_times.sh r8 >> $O
