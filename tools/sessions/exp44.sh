#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp44_late_y.txt
echo "# accumulating pass adds its y at the end (cache policy bit 7) against starting from it" > $O
DNM_CACHE_POLICY=226 timeout 900 python3 -m pytest tests/test_gpu_matvec.py -x -q -m gpu 2>&1 | tail -2 >> $O
one() { timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2 3 4; do
  echo "late-y" >> $O; DNM_CACHE_POLICY=226 one >> $O
  echo "default" >> $O; one >> $O
done
bash tools/pass_times.sh late DNM_CACHE_POLICY=226 >> $O
bash tools/pass_times.sh def >> $O
