#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp30_transpose_probe.txt
echo "# compute share of the transposed exchange, one rank at full size on one GPU" > $O
for s in 14 15; do DNM_SWZ=$s timeout 600 python3 tools/transpose_probe.py 30 8 5 2>&1 | grep -v amdgpu >> $O; done
DNM_SWZ=14 timeout 600 python3 tools/transpose_probe.py 30 4 1 2>&1 | grep -v amdgpu >> $O
echo "# swizzle shift at L=30, one GPU" >> $O
for s in 14 13 16 14 13; do
  echo "DNM_SWZ=$s" >> $O
  DNM_SWZ=$s timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])" >> $O
done
