#!/bin/bash
# usage: tools/pass_times.sh TAG [env assignments...] -- per-launch times of the tiled passes in a short bench.py run
# (rocprofv3 kernel trace; the program itself follows `--`, environment is exported beforehand)
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pt_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -f csv -d $OUT -o t -- python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 2 > $OUT/run.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "tile_pass" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = collections.OrderedDict()
    for i, r in enumerate(rows[4:]):            # skip the warm-up launches; the passes of a two-launch plan alternate
        per.setdefault(r["Kernel_Name"].split("(")[0][-40:] + " pass %d" % (i % 2), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    tot = 0.0
    for k, v in per.items():
        tot += sum(v) / len(v)
        print("$TAG", k, "n=%d avg %.3f ms min %.3f" % (len(v), sum(v) / len(v), min(v)))
    print("$TAG", "sum of pass averages %.3f ms" % tot)
PY
