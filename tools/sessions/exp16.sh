#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 900 bash tools/profile_bench.sh > gpurun_out/r02/profile_bench.log 2>&1
cp gpurun_out/profiles/kernel_stats.csv gpurun_out/r02/r02_bench_kernel_stats.csv
cp gpurun_out/profiles/kernel_trace_tail.csv gpurun_out/r02/r02_bench_kernel_trace_tail.csv
cp gpurun_out/profiles/pmc_summary.json gpurun_out/r02/r02_bench_pmc_summary.json
cp gpurun_out/profiles/bench_line.json gpurun_out/r02/r02_bench_line_under_rocprof.json
timeout 600 python3 bench.py > gpurun_out/r02/r02_bench_default_line.json 2> gpurun_out/r02/bench_stderr.txt
timeout 600 python3 tools/size_scan.py > gpurun_out/r02/r02_size_scan.txt 2>&1
tail -5 gpurun_out/r02/profile_bench.log
cat gpurun_out/r02/r02_bench_default_line.json
