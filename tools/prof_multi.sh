#!/bin/bash
# usage: tools/prof_multi.sh L 'cfgjson' ['ENV=V ...'] ... : for each config prints per-pass ms, FETCH (x2 corrected) and WRITE bytes per amplitude
set -u
L=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
i=0
for CFG in "$@"; do
  i=$((i+1))
  OUT=/tmp/prof_$i; rm -rf $OUT; mkdir -p $OUT
  ENVS=$(python3 -c "import json,sys; c=json.loads(sys.argv[1]); print(' '.join('%s=%s'%(k,v) for k,v in c.get('env',{}).items()))" "$CFG")
  export SWEEP="[$CFG]"
  env $ENVS rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc1 -o p -- python3 tools/sweep.py $L > $OUT/run1.txt 2>&1
  env $ENVS rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc2 -o p -- python3 tools/sweep.py $L > /dev/null 2>&1
  python3 - "$OUT" "$L" "$CFG" <<'PY'
import sqlite3, sys, glob, json
out, L, cfg = sys.argv[1], int(sys.argv[2]), sys.argv[3]
dim = 1 << L
def rows(db):
    c = sqlite3.connect(glob.glob(db)[0])
    return list(c.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like '%tile_pass%' order by dispatch_id"))
r1 = rows(out + "/pmc1/*.db"); r2 = rows(out + "/pmc2/*.db")
n = json.loads(cfg)
# number of launches per multiply = count distinct dispatches in last multiply: use durations pattern
d1 = {}
for d, name, v, dur in r1: d1[d] = (v, dur)
d2 = {}
for d, name, v, dur in r2: d2.setdefault(d, {})[name] = v
ids = sorted(d1)
# find period: launches per step from run1.txt
import re
m = re.search(r"launches=(\d+)", open(out + "/run1.txt").read())
per = int(m.group(1)) if m else 1
last = ids[-per:]
tot_ms = 0; tot_f = 0; tot_w = 0
line = []
for d in last:
    f = d1[d][0] * 1024 * 2 / dim; ms = d1[d][1] / 1e6
    w = d2.get(d, {}).get("WRITE_SIZE", 0) * 1024 / dim
    h = d2.get(d, {}).get("TCC_HIT_sum", 0); mi = d2.get(d, {}).get("TCC_MISS_sum", 0)
    line.append("[%.2f ms F=%.1f W=%.1f B/amp hit=%.0f%%]" % (ms, f, w, 100 * h / max(1, h + mi)))
    tot_ms += ms; tot_f += f; tot_w += w
print(cfg, " ".join(line), "TOTAL %.2f ms F+W=%.1f B/amp" % (tot_ms, tot_f + tot_w), flush=True)
PY
done
