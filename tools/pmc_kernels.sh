#!/bin/bash
# usage: tools/pmc_kernels.sh KERNEL_SUBSTR 'CTR ...' -- cmd...   last dispatch of EVERY kernel whose name matches
set -u
K=$1; CTRS=$2; shift 3
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/pmc_k; rocprofv3 --pmc $CTRS -d /tmp/pmc_k -o p -- "$@" > /tmp/pmc_k.txt 2>&1
python3 - "$K" <<'PY'
import sqlite3, sys, glob
c = sqlite3.connect(glob.glob("/tmp/pmc_k/*.db")[0])
rows = list(c.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size from counters_collection where kernel_name like ? order by dispatch_id", ("%" + sys.argv[1] + "%",)))
last = {}
for d, k, n, v, dur, g in rows:
    last[k] = max(last.get(k, -1), d)
for d, k, n, v, dur, g in rows:
    if d == last[k]:
        short = k.split("(")[0][-60:]
        print("   %-60s %-14s %18.0f   (%.3f ms, grid %d)" % (short, n, v, dur / 1e6, g))
PY
