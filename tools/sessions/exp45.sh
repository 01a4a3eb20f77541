#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp45_after_order.txt
echo "# with the window pass first and y added late: where the diagonal goes, rows per thread" > $O
one() { timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2 3; do
  echo "default" >> $O; one >> $O
  echo "diag-last" >> $O; DNM_DIAG_PASS=last one >> $O
  echo "rows=4" >> $O; DNM_LOG_ROWS=2 one >> $O
  echo "rows=4,diag-last" >> $O; DNM_LOG_ROWS=2 DNM_DIAG_PASS=last one >> $O
done
bash tools/pass_times.sh def >> $O
bash tools/pass_times.sh dl DNM_DIAG_PASS=last >> $O
