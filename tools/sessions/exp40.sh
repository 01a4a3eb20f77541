#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp40_window_first.txt
echo "# window pass first (writes y), contiguous pass second (accumulates): the bandwidth-bound pass loses its y read" > $O
one() { timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['config']['plan'][60:400])"; }
for i in 1 2; do
  echo "window first" >> $O; DNM_WINDOW_FIRST=1 one >> $O
  echo "default" >> $O; one >> $O
done
bash tools/pass_times.sh wf DNM_WINDOW_FIRST=1 >> $O
bash tools/pass_times.sh def >> $O
