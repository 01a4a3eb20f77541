#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp29_transpose.txt
echo "# transposed exchange: staged multi-rank tests on one GPU" > $O
timeout 1500 python3 -m pytest tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -15 >> $O
echo "# S=15 / S=14 against S=16 at L=30 (one GPU)" >> $O
for s in 16 15 14 16 15; do
  echo "DNM_SWZ=$s" >> $O
  DNM_SWZ=$s timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])" >> $O
done
