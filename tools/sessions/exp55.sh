#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp55_sched_rows4.txt
echo "# scheduler options for the kernel file again, now that 4 rows per thread at 64 registers is the default" > $O
one() { timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3 4; do
  for v in default plain relaxed maxilp memclause; do
    echo -n "$v " >> $O
    if [ $v = default ]; then unset DNM_LIB; else export DNM_LIB=$PWD/dynamite_amd/build/lib_$v.so; fi
    one >> $O
  done
done
