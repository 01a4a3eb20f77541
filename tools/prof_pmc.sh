#!/bin/bash
# usage: tools/prof_pmc.sh L 'cfgjson' 'CTR CTR ...' ['CTR ...' ...] : one rocprofv3 --pmc pass per counter group; prints last tile_pass dispatches
set -u
L=$1; CFG=$2; shift 2
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
ENVS=$(python3 -c "import json,sys; c=json.loads(sys.argv[1]); print(' '.join('%s=%s'%(k,v) for k,v in c.get('env',{}).items()))" "$CFG")
export SWEEP="[$CFG]"
echo "== $CFG"
i=0
for GRP in "$@"; do
  i=$((i+1)); OUT=/tmp/pmc_$i; rm -rf $OUT
  env $ENVS rocprofv3 --pmc $GRP -d $OUT -o p -- python3 tools/sweep.py $L > $OUT.txt 2>&1
  python3 - "$OUT" <<'PY'
import sqlite3, sys, glob
c = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
rows = list(c.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like '%tile_pass%' order by dispatch_id"))
last = max(r[0] for r in rows)
for d, n, v, dur in rows:
    if d == last: print("   %-36s %16.0f   (%.2f ms)" % (n, v, dur / 1e6))
PY
done
