#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp46_rows4_series.txt
echo "# longer alternating series: rows per thread x diagonal placement (window pass first, y added late)" > $O
one() { timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3 4 5 6 7 8; do
  echo -n "A rows=8 " >> $O; one >> $O
  echo -n "B rows=4 " >> $O; DNM_LOG_ROWS=2 one >> $O
  echo -n "C rows=4,diag-last " >> $O; DNM_LOG_ROWS=2 DNM_DIAG_PASS=last one >> $O
done
