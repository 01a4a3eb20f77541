#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp41_pass_order.txt
echo "# order of the two passes x which of them carries the diagonal; alternating processes, same box" > $O
one() { timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2 3 4; do
  echo "window-first diag-first" >> $O; DNM_WINDOW_FIRST=1 one >> $O
  echo "window-first diag-last" >> $O; DNM_WINDOW_FIRST=1 DNM_DIAG_PASS=last one >> $O
  echo "default(contiguous-first diag-first)" >> $O; one >> $O
done
bash tools/pass_times.sh wf_dl DNM_WINDOW_FIRST=1 DNM_DIAG_PASS=last >> $O
bash tools/pass_times.sh wf_df DNM_WINDOW_FIRST=1 >> $O
