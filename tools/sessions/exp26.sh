#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
SECONDS=0; timeout 900 python3 bench.py > gpurun_out/r02/r02_bench_default_line.json 2> gpurun_out/r02/bench_time.txt
echo "bench wall seconds: $SECONDS"
cat gpurun_out/r02/r02_bench_default_line.json | cut -c1-300
python3 -c "
import json; d=json.loads(open('gpurun_out/r02/r02_bench_default_line.json').read().strip().splitlines()[-1]); print(d['cpu_baseline'])"
