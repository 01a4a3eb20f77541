#!/bin/bash
# GPU box: the SYK multiply with table records (DNM_TAB_RECORDS=1, default) and with records of four
# terms (=0): ms per multiply, then counters summed over the passes of ONE multiply (the last NP tile_pass dispatches).
#   tools/probes/syk_counters.sh [L=24] [NP=23] [variants="1 0"]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp DNM_EXPERIMENTAL=1
L=${1:-24}; NP=${2:-23}; VARS=${3:-"1 0"}
for t in $VARS; do
  export DNM_TAB_RECORDS=$t
  echo "######## DNM_TAB_RECORDS=$t DNM_TAB_LOG_ROWS=${DNM_TAB_LOG_ROWS:-default}"
  python3 tools/models_bench.py syk:full:$L 2>&1 | grep "multiply"
  for set in 'SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS' \
             'TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TD_TD_BUSY_sum' 'FETCH_SIZE TCC_HIT_sum TCC_MISS_sum'; do
    rm -rf /tmp/pmc_s; rocprofv3 --pmc $set -d /tmp/pmc_s -o p -- python3 tools/models_bench.py syk:full:$L > /tmp/pmc_s.txt 2>&1
    python3 - $NP <<'PY'
import sqlite3, sys, glob
c = sqlite3.connect(glob.glob("/tmp/pmc_s/*.db")[0])
rows = list(c.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like '%tile_pass%' order by dispatch_id"))
ids = sorted({r[0] for r in rows})[-int(sys.argv[1]):]
tot, ms = {}, {}
for d, n, v, dur in rows:
    if d in ids:
        tot[n] = tot.get(n, 0) + v
        ms[n] = ms.get(n, 0) + dur / 1e6
for n in tot:
    print("   %-22s %20.0f   (summed over %d passes, %.1f ms of kernels under the profiler)" % (n, tot[n], len(ids), ms[n]))
PY
  done
done
