#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp31_prefetch.txt
echo "# first gathered record issued behind the tile loads; accumulator loads before the LDS stores; one init mode and a fixed cache policy per kernel" > $O
timeout 900 python3 -m pytest tests/test_gpu_matvec.py -x -q -m gpu 2>&1 | tail -3 >> $O
one() { timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2 3; do
  echo "new" >> $O; one >> $O
  echo "prev" >> $O; DNM_LIB=$PWD/dynamite_amd/build/lib_prev.so one >> $O
done
