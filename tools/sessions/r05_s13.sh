#!/bin/bash
# round 5, GPU session 13: what the real-arithmetic bond-graph passes cost (kernel times + SQ counters), full suite
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s13; mkdir -p $OUT
M=$OUT/kagome_real.txt
python3 tools/models_bench.py --real kagome30:sc kagome30:scx 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-200 | tee $M
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- python3 tools/models_bench.py --real kagome30:sc > /dev/null 2>&1
python3 - <<'PY' | tee -a $M
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "sc3" in r["Name"]:
        print("   %-70s calls %5s  avg %9.3f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  echo "-- kagome30:sc real: $G" | tee -a $M
  bash tools/pmc_kernels.sh sc3g "$G" -- python3 tools/models_bench.py --real kagome30:sc | tee -a $M
done
timeout 1500 python3 -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -22 > $OUT/suite.txt; tail -3 $OUT/suite.txt
